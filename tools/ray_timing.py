"""Measurement aid: time stamps of one workgroup of the segmented ray kernel (build variant with
-DGIE_RAY_TIMING=<block>; the stamps overwrite the start of the edt plane)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK = sys.argv[2] if len(sys.argv) > 2 else "200"
LIB = os.path.join(ROOT, "tools", "ablate", "libgie_hip_rt%s.so" % BLOCK)
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           "-DGIE_RAY_TIMING=" + BLOCK, os.path.join(ROOT, "gie-mapping_amd", "csrc", "gie_hip.hip"), "-o", LIB])
    sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import mapper, scenes
mapper.load_library(LIB)
frames = bench.make_frames(scenes, 0.05, 4, 5, "vlp16")
m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
for pos, q, pts, _ in frames[:3]:
    m.update(pos, q, "pointcloud", pts)
pos, q, pts, _ = frames[3]
m.set_pose(pos, q); m.ogm_pointcloud(pts); m.sync()
e = m.read_local(vtype=False, dist_sq=False, coc=False)["edt"].ravel()[:16 * 8].reshape(16, 8)
t0 = e[:8, 0].min()
print("seg  start  state+publish  phase1  barrier  phase2   (us since the first stamp)")
for s in range(8):
    print("%3d %6.1f %10.1f %10.1f %8.1f %8.1f" % ((s,) + tuple(((e[s, k] - t0) % 16777216) / 100.0 for k in range(5))))
