#!/bin/bash
# measurement aid (GPU box): the C5 update for a list of persistent wavefront grid sizes (gie_config.wave_workgroups through bench.py)
for w in "$@"; do
  W=$w GIE_BENCH_WAVE_WGS=$w python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print('wgs', os.environ.get('W'), d['ms_per_step'], d['kernels_ms_per_step']['waves'])" 
done
