"""Measurement aid (under rocprofv3 --kernel-trace): one 512^3 tile of a 2x2x2 arrangement, K map
updates, each followed (argv[1] = 1) or not (0) by the device side of one exchange round with
three shared faces (export_all -> its own layers back as ghosts -> refine): which kernels does a
round add, and how long are they?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]
import numpy as np, torch, bench, gie
from gie import scenes, tiling
ex = int(sys.argv[1])
size = (512, 512, 512); world = 8; rank = 0
frames = bench.make_frames(scenes, 0.05, 13, 5, "vlp16")
dev = torch.device("cuda", 0)
d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
grid = tiling.tile_grid(world); whole = tuple(grid[i] * size[i] for i in range(3))
m = gie.Mapper(gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False))
m.set_tile(tiling.tile_offset_voxels(rank, world, size), whole)
nbs = tiling.neighbours(rank, world)
bufs = {f: torch.empty(m.halo_count(f) * 20, dtype=torch.uint8, device=dev) for f in nbs}
import time
for i, (pos, q, pts, _) in enumerate(frames):
    if i == 3:
        m.sync(); t0 = time.perf_counter()
    m.set_pose(pos, q); m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0]); m.step_begin_tiled()
    if ex:
        for rnd in range(2):                              # round 0 finishes the merge, round 1 refines
            m.halo_export_all_dev({f: bufs[f].data_ptr() for f in nbs})
            m.halo_import_all_dev({f: bufs[f].data_ptr() for f in nbs})
            if rnd == 0:
                m.merge_end()
            else:
                m.refine_async()
    else:
        m.merge_end()
m.sync()
print("exchange", ex, "ms per update %.4f" % (1e3 * (time.perf_counter() - t0) / 10))
