#!/bin/bash
# Copies the summaries tools/profile_final.sh left under gpurun_out/prof_<R>_<workload>/ into profiles/ (tracked) and merges their
# traffic entries into the file bench.py's roofline.traffic reads.   tools/collect_profiles.sh r06
set -u
R=${1:-r06}
ARGS=""
for d in gpurun_out/prof_${R}_*; do
  [ -d "$d" ] || continue
  wl=${d#gpurun_out/prof_${R}_}
  [ -f $d/kernel_stats.txt ] && cp $d/kernel_stats.txt profiles/${R}_${wl}_kernel_stats.txt
  [ -f $d/hbm_traffic.txt ] && cp $d/hbm_traffic.txt profiles/${R}_${wl}_hbm_traffic.txt
  if [ -f $d/traffic_entry.json ]; then
    size=$(python - "$wl" <<'PY'
import sys
sys.path.insert(0, ".")
import bench
s = bench.PRESETS[sys.argv[1]]["size"]
print("%dx%dx%d" % tuple(s))
PY
)
    ARGS="$ARGS ${wl}_${size}=$d/traffic_entry.json"
  fi
done
[ -f gpurun_out/${R}_c5_sq_counters.txt ] && cp gpurun_out/${R}_c5_sq_counters.txt profiles/${R}_c5_sq_counters.txt
python tools/merge_traffic.py $ARGS
