#!/usr/bin/env python3
"""Short view of a bench.py JSON line: python tools/bench_summary.py gpurun_out/bench.json"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["preset"], d["value"], "Mvox/s", d["ms_per_step"], "ms; step", d["step_ms"]["median"], "p95", d["step_ms"]["p95"])
print("  ", json.dumps(d["kernels_ms_per_step"]))
print("   roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "sweep", d["roofline_wavefront_sweep"]["frac"], "update", d["roofline_update"]["frac"])
for k, v in d.get("extra_runs", {}).items():
    print(k, v["ms_per_step"], "ms; step", v["step_ms"]["median"])
    print("  ", json.dumps(v["kernels_ms_per_step"]))
