"""Measurement aid: time stamps at every phase boundary of the waves kernel (build variant with -DGIE_WAVE_TIMING; the boss
thread writes them into the middle row of the edt plane).   python tools/wave_timing.py build | run [workload]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "ablate", "libgie_hip_wt.so")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           "-DGIE_WAVE_TIMING=1", os.path.join(ROOT, "gie-mapping_amd", "csrc", "gie_hip.hip"), "-o", LIB])
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import mapper, scenes
mapper.load_library(LIB)
wl = sys.argv[2] if len(sys.argv) > 2 else "c5"
dev = torch.device("cuda", 0)
size = (512, 512, 512)
m = gie.Mapper(gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False))
feed = bench.make_feed(wl, torch, scenes, dev, 0.05, size, (0, 0, 0), 8)
feed.prepare(0, 8)
for i in range(8):
    feed.step_input(m, i); m.step(); m.sync()
st = m.stats()
row = m.read_local(vtype=False, dist_sq=False, coc=False)["edt"][size[2] // 2, size[1] // 2, :]
names = {0: "start", 1: "A level start", 2: "A phase 1 + barrier", 3: "A phase 2 + barrier", 4: "B start", 5: "B phase 1 (first) + barrier", 6: "B phase 2 + barrier",
         7: "B phase 3 + next phase 1 + barrier", 8: "A done + grid barrier", 9: "B done + grid barrier", 10: "C done"}
print("visits", st["visits_a"], st["visits_b"], st["visits_c"], "levels", st["levels_a"], st["levels_b"], st["levels_c"])
prev = None
tot = {}
for k in range(0, 1000, 2):
    t, tag = float(row[k]), int(row[k + 1])
    if k > 0 and tag // 1000000 == 0 and t == 0:
        break
    tg, n = tag // 1000000, tag % 1000000
    if prev is not None:
        dt = (t - prev) % 16777216.0 / 100.0
        print("%-38s n %7d   %8.2f us" % (names.get(tg, tg), n, dt))
        tot[tg] = tot.get(tg, 0.0) + dt
    prev = t
    if tg == 10:
        break
print({names[k]: round(v, 1) for k, v in tot.items()})
