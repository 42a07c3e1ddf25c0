"""Evidence run (GPU box, under rocprofv3 --kernel-trace --stats): what bringing EVERY deferred record up to date costs at 512^3
(gie_catchup_everything: k_coc_catchup_list + k_coc_catchup_run) — the price a node pays once when it switches the changed-block
stream on (gie_stream_enable) while 82 % of the headline volume's tiles have their stored records in the pair plane only.
    rocprofv3 --kernel-trace --stats -d out -o t -- python tools/catchup_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
import gie  # noqa: E402
from gie import scenes  # noqa: E402

size = (512, 512, 512)
dev = torch.device("cuda", 0)
m = gie.Mapper(gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False, wave_workgroups=160))
feed = bench.make_feed("c5", torch, scenes, dev, 0.05, size, (0, 0, 0), 16)
feed.prepare(0, 12)
for i in range(12):
    on = i in (6, 10)                 # two updates in the reference's order of kernels, four fused ones in between
    m.stream_enable(on)
    m.sync(); t0 = time.perf_counter()
    feed.step_input(m, i); m.step(); m.sync()
    print("update %2d  %s  %.3f ms (host clock, whole update)" % (i, "stream on (catch-up of everything + Mark ... commit as separate sweeps)" if on else "fused", 1e3 * (time.perf_counter() - t0)), flush=True)
m.close()
