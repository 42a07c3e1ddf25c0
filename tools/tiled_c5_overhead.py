#!/usr/bin/env python3
"""What the halo exchange of a tiled C5 run costs on the DEVICE, measured on one GPU: 2x2x2 tiles (default 256^3 each) of the
hash world as eight mappers of this process, (a) every tile on its own (plain map updates), (b) the tiled sequence — split
merge, export / import of the face layers between the mappers (device-resident, stream-ordered), `rounds` refinement rounds.
All tiles share the one GPU, so the times are sums over the eight tiles; (b) / (a) is the per-rank overhead a multi-GPU run pays
on top of the transfers.   python tools/tiled_c5_overhead.py [tile] [rounds] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))
import torch  # noqa: E402
import gie  # noqa: E402
from gie import scenes, tiling  # noqa: E402
import bench  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tile = (T, T, T)
dev = torch.device("cuda", 0)
grid = tiling.tile_grid(8)
whole = tuple(grid[i] * tile[i] for i in range(3))
cfg = gie.make_config(0.05, tile, cutoff_dist=2.0, fast_mode=False)


def run(tiled):
    ms, feeds = [], []
    for r in range(8):
        m = gie.Mapper(cfg)
        off = tiling.tile_offset_voxels(r, 8, tile)
        if tiled:
            m.set_tile(off, whole)
        ms.append(m)
        f = bench.HashWorldFeed(torch, scenes, dev, 0.05, tile, off if tiled else (0, 0, 0))
        feeds.append(f)
    for f in feeds:
        f.prepare(0, 3 + K)
    bufs = {}
    stats = None
    t0 = None
    for i in range(3 + K):
        if i == 3:
            for m in ms:
                m.sync()
            t0 = time.perf_counter()
        for m, f in zip(ms, feeds):
            f.step_input(m, i)
            if tiled:
                m.step_begin_tiled()
            else:
                m.step()
        if tiled:
            tiling.exchange_rounds_local_device(ms, grid, dev, rounds=rounds, bufs=bufs)
    for m in ms:
        m.sync()
    dt = 1e3 * (time.perf_counter() - t0) / K
    stats = [m.stats() for m in ms]
    for st in stats:
        st["visits_c_per_update"] = st["total_visits_c"] / float(3 + K)
        st["visits_ab_per_update"] = (st["total_visits_a"] + st["total_visits_b"]) / float(3 + K)
    for m in ms:
        m.close()
    return dt, stats


a, sa = run(False)
b, sb = run(True)
print("tiles of %d^3, %d refinement round(s): 8 independent tiles %.3f ms per update of all eight; tiled %.3f ms; ratio %.3f" % (T, rounds, a, b, b / a))
print("wave C visits per tile and map update (all wavefront launches of an update, mean over the run): independent %s | tiled %s" % (
    [int(s_["visits_c_per_update"]) for s_ in sa], [int(s_["visits_c_per_update"]) for s_ in sb]))
print("wave A + B visits per tile and map update: independent %s | tiled %s" % ([int(s_["visits_ab_per_update"]) for s_ in sa], [int(s_["visits_ab_per_update"]) for s_ in sb]))
print("wave C visits per tile and update, last update: independent %s | tiled %s" % ([s["visits_c"] for s in sa], [s["visits_c"] for s in sb]))
