"""Measurement aid: per-phase time stamps of wave C (build variant with -DGIE_WAVE_TIMING; the
stamps overwrite the start of the edt plane).  tools/wave_timing.py build | run"""
import math, os, subprocess, sys
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
sensor = "vlp16_projective"
rings, az, phi_min, phi_inc, bins = bench.SENSORS[sensor]
frames = bench.make_frames(scenes, 0.05, 8, 5, sensor)
dev = torch.device("cuda", 0)
d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
for i, (pos, q, pts, _) in enumerate(frames):
    m.set_pose(pos, q)
    m.ogm_multiscan_dev(d_pts[i].data_ptr(), bins, rings, 2.0 * math.pi / bins, -math.pi, math.radians(phi_inc), math.radians(phi_min))
    m.fuse(); m.batch_edt()
    if i == 7:
        # stop before commit overwrites the stamps: run merge's pieces through gie_merge, then read edt (commit only writes known voxels' edt)
        pass
    m.merge(); m.sync()
st = m.stats()
e = m.read_local(vtype=False, dist_sq=False, coc=False)["edt"].ravel()[:8 * 400].reshape(-1, 8)
lv = st["levels_c"]
print("levels", lv, "visits", st["visits_c"])
d = np.diff(e[:, :7], axis=1)
d = np.where(d < 0, d + 16777216.0, d)          # 24-bit wrap
ok = (e[:, 0] > 0) & (np.abs(d) < 1e5).all(1)
names = ["relax", "ballot+sync", "reserve+sync", "stores+sync", "grid barrier", "read n"]
print("rows used", int(ok.sum()))
for k, nme in enumerate(names):
    print("%-14s %.2f us" % (nme, d[ok, k].mean() / 100.0))
print("sum %.2f us" % (d[ok].sum(1).mean() / 100.0))
