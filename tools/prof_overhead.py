import sys, time, math, os
sys.path[:0] = ["/root/repo", "/root/repo/gie-mapping_amd"]
import torch, bench, gie
from gie import scenes
sensor = "vlp16"
rings, az, phi_min, phi_inc, bins = bench.SENSORS[sensor]
frames = bench.make_frames(scenes, 0.05, 43, 5, sensor)
dev = torch.device("cuda", 0)
d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
for prof in (0, 1, 0, 1):
    m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
    def step(i):
        m.set_pose(frames[i][0], frames[i][1]); m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0]); m.step()
    for i in range(3): step(i)
    m.sync(); m.profile_enable(bool(prof)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3, 43): step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m.sync(); print("profiling", prof, "ms/step %.4f" % (1e3 * dt / 40)); m.close()
