#!/bin/bash
# Runs ON THE GPU BOX: everything the round commits under profiles/ besides the per-workload profiles of tools/profile_final.sh.
#   tools/evidence_round.sh r06
set -u
R=${1:-r06}
ROOT=$(pwd); O=$ROOT/gpurun_out/ev_$R; mkdir -p $O
export TMPDIR=/tmp
for wl in c5 vlp16_projective c3_projective; do
  n=8; [ $wl != c5 ] && n=14
  { echo "# GIE_EDT_RAW=1 python tools/wave_timing.py run $wl $n v   (measurement build -DGIE_WAVE_TIMING: the clocks cost time themselves)"; GIE_EDT_RAW=1 timeout 400 python tools/wave_timing.py run $wl $n v 2>&1 | grep -E "^frame|blocks:|tiles:|inside the levels"; } > $O/wave_timing_$wl.txt
done
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/cu -o cu -- python $ROOT/tools/catchup_time.py > $O/catchup_run.txt 2> $O/catchup.err )
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/catchup_time.py   (512^3 headline world; updates 6 and 10 run with gie_stream_enable: gie_catchup_everything)"; cat $O/catchup_run.txt; python tools/rocpd_summary.py stats $(find $O/cu -name "*.db" | head -1) | grep -E "^kernel|catchup|mark|commit|oldskip"; } > $O/catchup_everything_512.txt
rm -rf $O/cu
python tools/soak_fullsize.py --c5 --frames 8 > $O/soak_fullsize_c5_8_updates.log 2>&1
python tools/soak_fullsize.py --frames 9 --pattern rpp > $O/soak_fullsize_9_updates_rpp.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
cp profiles/bench_last_full.json $O/bench_driver_cmd_full.json
ls -la $O
