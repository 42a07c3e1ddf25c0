"""Measurement aid: is a map update bound by the GPU or by the host's enqueueing?  Times the loop
that enqueues K map updates (no synchronisation inside) and the wait for the GPU afterwards."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import torch, bench, gie
from gie import scenes
K = 40
frames = bench.make_frames(scenes, 0.05, K + 3, 5, "vlp16")
dev = torch.device("cuda", 0)
d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
m = gie.Mapper(gie.make_config(0.05, (512, 512, 512), cutoff_dist=2.0, fast_mode=False))
def step(i):
    m.set_pose(frames[i][0], frames[i][1]); m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0]); m.step()
for i in range(3): step(i)
m.sync(); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 3 + K): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.4f ms per map update, then %.4f ms of waiting per map update (total %.4f)" % (1e3 * (t1 - t0) / K, 1e3 * (t2 - t1) / K, 1e3 * (t2 - t0) / K))
