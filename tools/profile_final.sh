#!/bin/bash
# Runs ON THE GPU BOX: the round's committed evidence.  tools/profile_final.sh r06
#   C5 (the driver's command without extras): kernel stats + HBM traffic (PMC, separate passes) + SQ counters
#   lidar workloads and BASELINE configs 2 / 3 / 4: kernel stats + HBM traffic
set -u
R=${1:-r06}
ROOT=$(pwd)
bash tools/profile_round.sh ${R}_c5 "--gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline" > /dev/null 2>&1
bash tools/pmc_run.sh > gpurun_out/${R}_c5_sq_counters.txt 2>&1
for wl in vlp16_projective vlp16 c2 c2_projective c3 c3_projective c4 c4_nofast; do
  bash tools/profile_round.sh ${R}_$wl "--workload $wl --steps 10 --warmup 3 --no-extras --no-cpu-baseline" > /dev/null 2>&1
done
ls gpurun_out | grep prof_${R}
