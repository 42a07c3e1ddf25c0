#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the rocprofv3 evidence behind bench.py's roofline object.
#   pass 1  --kernel-trace --stats   (per-kernel durations of the same command)
#   pass 2  --pmc FETCH_SIZE         (own run, kernel-trace only: PMC passes never share a run with other traces)
#   pass 3  --pmc WRITE_SIZE
# Summaries land in gpurun_out/prof_<tag>/ ; copy what is to be judged into profiles/.
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-timed-s 0"}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
if [ "${TRACE_ONLY:-0}" = "1" ]; then
  T=$(find $OUT/trace -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (MI355X)"; python $ROOT/tools/rocpd_summary.py stats "$T"; } > $OUT/kernel_stats.txt
  rm -rf $OUT/trace; cat $OUT/kernel_stats.txt; exit 0
fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o write -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/write.err
cd $ROOT
T=$(find $OUT/trace -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py $ARGS   (MI355X)"; python tools/rocpd_summary.py stats "$T"; echo; echo "# bench.py line of the same (profiled) run:"; cat $OUT/bench_trace.json; } > $OUT/kernel_stats.txt
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and  --pmc WRITE_SIZE (two separate passes) -- python bench.py $ARGS"; python tools/rocpd_summary.py pmc "$F" "$W" $OUT/traffic.json; } > $OUT/hbm_traffic.txt
# the entry bench.py's roofline.traffic reads: stamped with the content hash of the kernel sources (bench.csrc_hash) so that it is
# only ever priced against durations of the same kernels; merge it into profiles/traffic_r06.json with tools/merge_traffic.py
python - "$OUT/traffic.json" "$TAG" "$ARGS" > $OUT/traffic_entry.json <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
t = json.load(open(sys.argv[1]))
print(json.dumps({"source": "profiles/%s_hbm_traffic.txt: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python bench.py %s`, "
                            "%d map updates, fetch x2 (gfx950 wide-stream correction), bytes per map update" % (sys.argv[2], sys.argv[3], t.get("_map_updates", 0)),
                  "csrc_hash": bench.csrc_hash(),
                  "kernels": t.get("_per_step_by_stage", {})}, indent=1))
PY
rm -rf $OUT/trace $OUT/fetch $OUT/write      # the databases are large; the summaries are what travels back
ls -la $OUT
