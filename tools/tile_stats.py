"""How much of the volume do the sweeps really have to look at?  Known-tile statistics of the bench workloads."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import scenes
for sensor in ("vlp16", "vlp16_projective"):
    rings, az, phi_min, phi_inc, bins = bench.SENSORS[sensor]
    frames = bench.make_frames(scenes, 0.05, 8, 5, sensor)
    dev = torch.device("cuda", 0)
    d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
    m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
    for i, (pos, q, pts, _) in enumerate(frames):
        m.set_pose(pos, q)
        if bins is None:
            m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0])
        else:
            m.ogm_multiscan_dev(d_pts[i].data_ptr(), bins, rings, 2.0 * math.pi / bins, -math.pi, math.radians(phi_inc), math.radians(phi_min))
        m.step()
    ty = m.read_local(edt=False, dist_sq=False, coc=False)["type"]
    kn = (ty != 0).reshape(64, 8, 64, 8, 64, 8).any(axis=(1, 3, 5))          # [tz, ty, tx]
    bd = np.zeros_like(kn); bd[0] = bd[-1] = True; bd[:, 0] = bd[:, -1] = True; bd[:, :, 0] = bd[:, :, -1] = True
    need = kn | bd
    cols = need.any(axis=0)
    print(sensor, "known voxels %.4f" % (ty != 0).mean(), "known tiles %.3f" % kn.mean(), "need tiles (known|boundary) %.3f" % need.mean(),
          "xy tile columns with any need %.3f" % cols.mean(), "occupied planes", int((ty == 2).any(axis=(1, 2)).sum()))
    m.close()
