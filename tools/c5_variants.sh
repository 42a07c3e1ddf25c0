#!/bin/bash
# Runs ON THE GPU BOX: the C5 headline workload under a list of environment variants, one bench line each.
#   tools/c5_variants.sh "NAME=VALUE ..." "NAME=VALUE ..." ...     ("" = defaults)
# The GIE_* switches only exist in the TEST build of the library (tests/gpu_helpers/libgie_hip_test.so, -DGIE_TEST_HOOKS): the
# package loads it through GIE_LIB; the product library reads no environment.
set -u
export GIE_LIB=$(pwd)/tests/gpu_helpers/libgie_hip_test.so
ROOT=$(pwd); OUT=$ROOT/gpurun_out/variants; mkdir -p $OUT
i=0
for v in "$@"; do
  i=$((i+1))
  echo "== variant $i: $v"
  env $v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --min-timed-s 0.2 > $OUT/v$i.json 2> $OUT/v$i.err || tail -3 $OUT/v$i.err
  python - "$OUT/v$i.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("  ms_per_step %.4f  value %.0f  kernels %s" % (d["ms_per_step"], d["value"], json.dumps(d["kernels_ms_per_step"])))
    print("  wavefront %s  update %s" % (d["roofline_wavefront_sweep"]["frac"], d["roofline_update"]["frac"]))
except Exception as e:
    print("  no line:", e)
PY
done
