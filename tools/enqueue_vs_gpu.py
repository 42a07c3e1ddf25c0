#!/usr/bin/env python3
"""Host enqueue time vs GPU time of a map update: python tools/enqueue_vs_gpu.py [workload] [steps].
Prints the time the host needs to ENQUEUE a map update (no synchronisation inside the loop) next to the time per update
with the device drained at the end — when the first is not well below the second, the update is launch-bound."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))
import torch  # noqa: E402
import gie  # noqa: E402
from gie import scenes  # noqa: E402
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "vlp16"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
size = (512, 512, 512)
cfg = gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False)
feed = bench.make_feed(wl, torch, scenes, dev, 0.05, size, (0, 0, 0), K + 3)
m = gie.Mapper(cfg)
feed.prepare(0, 3)
for i in range(3):
    feed.step_input(m, i); m.step()
m.sync()
feed.prepare(3, K)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 3 + K):
    feed.step_input(m, i); m.step()
t1 = time.perf_counter()
m.sync()
t2 = time.perf_counter()
print("%s: host enqueue %.4f ms per update, with the device drained %.4f ms per update (%d updates)" % (wl, 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K, K))
# one update at a time (host waits for each): the latency of a single update
lat = []
feed.prepare(3 + K, 5)
for i in range(3 + K, 8 + K):
    t = time.perf_counter(); feed.step_input(m, i); m.step(); m.sync(); lat.append(1e3 * (time.perf_counter() - t))
print("   one update at a time: %s ms" % ", ".join("%.3f" % v for v in lat))
m.close()
