"""What does the halo exchange cost next to the map update?  Two 512^3 tiles (1024 x 512 x 512
volume) as two mappers on ONE GPU, device-resident exchange without any transport
(tiling.exchange_until_stable_local_device): the GPU-side work of a round (export, ghost import,
refinement), i.e. the floor a multi-GPU run adds RCCL transfers to."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import scenes, tiling

size = tuple(int(v) for v in (sys.argv[1:4] or (512, 512, 512)))
world = int(sys.argv[4]) if len(sys.argv) > 4 else 2
sensor = "vlp16"
rings, az, phi_min, phi_inc, bins = bench.SENSORS[sensor]
frames = bench.make_frames(scenes, 0.05, 10, 5, sensor)
dev = torch.device("cuda", 0)
d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
grid = tiling.tile_grid(world)
whole = tuple(grid[i] * size[i] for i in range(3))
ms = []
for r in range(world):
    m = gie.Mapper(gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False))
    m.set_tile(tiling.tile_offset_voxels(r, world, size), whole)
    ms.append(m)
upd, exc, rounds = [], [], []
for i, (pos, q, pts, _) in enumerate(frames):
    for m in ms: m.sync()
    t0 = time.perf_counter()
    for m in ms:
        m.set_pose(pos, q); m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0]); m.step_begin_tiled()
    for m in ms: m.sync()
    t1 = time.perf_counter()
    rounds.append(tiling.exchange_until_stable_local_device(ms, grid, dev))
    for m in ms: m.sync()
    t2 = time.perf_counter()
    upd.append(1e3 * (t1 - t0)); exc.append(1e3 * (t2 - t1))
# the same frames again, everything enqueued back to back: no exchange / one stream-ordered round per map update
def run(rounds):
    for m in ms: m.sync()
    t0 = time.perf_counter()
    for i, (pos, q, pts, _) in enumerate(frames):
        for m in ms:
            m.set_pose(pos, q); m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0]); m.step_begin_tiled()
        if rounds:
            tiling.exchange_rounds_local_device(ms, grid, dev, rounds=rounds)
    for m in ms: m.sync()
    return 1e3 * (time.perf_counter() - t0) / len(frames)
async_ms = {"no_exchange": round(run(0), 3), "one_round": round(run(1), 3), "two_rounds": round(run(2), 3)}
print(json.dumps({"tiles": world, "enqueued_back_to_back_ms_per_update_both_tiles": async_ms, "tile": size, "update_ms_both_tiles": [round(v, 3) for v in upd], "exchange_ms": [round(v, 3) for v in exc], "rounds": rounds}))
