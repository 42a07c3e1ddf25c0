import os, subprocess, sys
ROOT = "/root/repo"
LIB = os.path.join(ROOT, "tools", "ablate", "libgie_hip_rtall.so")
if sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC",
                           "-DGIE_RAY_TIMING=-1", os.path.join(ROOT, "gie-mapping_amd", "csrc", "gie_hip.hip"), "-o", LIB]); sys.exit(0)
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import mapper, scenes
mapper.load_library(LIB)
frames = bench.make_frames(scenes, 0.05, 4, 5, "vlp16")
m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
for pos, q, pts, _ in frames[:3]: m.update(pos, q, "pointcloud", pts)
pos, q, pts, _ = frames[3]
m.set_pose(pos, q); m.ogm_pointcloud(pts); m.sync()
nb = (len(pts) + 63) // 64
e = m.read_local(vtype=False, dist_sq=False, coc=False)["edt"].ravel()[:nb * 2].reshape(nb, 2).astype(np.float64)
t0 = e[:, 0].min()
st = ((e[:, 0] - t0) % 16777216) / 100.0; en = ((e[:, 1] - t0) % 16777216) / 100.0
print("blocks", nb, "start min/median/max %.1f %.1f %.1f us" % (st.min(), np.median(st), st.max()), "end max %.1f us" % en.max(), "duration median %.1f max %.1f" % (np.median(en - st), (en - st).max()))
print("starts histogram (10 us bins):", np.histogram(st, bins=np.arange(0, 110, 10))[0].tolist())
print("durations histogram (5 us bins):", np.histogram(en - st, bins=np.arange(0, 80, 5))[0].tolist())
o = np.argsort(en - st)[::-1][:8]
print("slowest blocks:", [(int(b), round(float(st[b]), 1), round(float(en[b] - st[b]), 1)) for b in o])
