"""Spatial sharding of a large local volume over the GPUs of one node.

A volume of `grid` voxels is cut into block-aligned tiles, one per rank; rank r owns the voxels
(and the global voxel blocks) of its tile.  Round 1 runs the tiles independently (no halo
exchange yet), which is what bench.py reports as weak scaling.
"""
import numpy as np


def tile_grid(world_size):
    """Tiles per axis (tx, ty, tz) for a world size that is a power of two: 1→(1,1,1), 2→(2,1,1),
    4→(2,2,1), 8→(2,2,2), …"""
    if world_size < 1 or world_size & (world_size - 1):
        raise ValueError("world_size must be a power of two")
    t = [1, 1, 1]
    a = 0
    while t[0] * t[1] * t[2] < world_size:
        t[a] *= 2
        a = (a + 1) % 3
    return tuple(t)


def tile_of_rank(rank, world_size, tile_size):
    """Origin (in voxels, relative to the large volume's corner) and size of rank's tile."""
    tx, ty, tz = tile_grid(world_size)
    ix, iy, iz = rank % tx, (rank // tx) % ty, rank // (tx * ty)
    size = tuple(int(s) for s in tile_size)
    if any(s % 8 for s in size):
        raise ValueError("tiles must be aligned to the 8-voxel blocks")
    return (ix * size[0], iy * size[1], iz * size[2]), size


def tile_centre_offset(rank, world_size, tile_size, voxel_width):
    """Metric offset of the tile centre from the centre of the whole volume: the pose a rank
    feeds its mapper is the shared sensor pose shifted by this."""
    origin, size = tile_of_rank(rank, world_size, tile_size)
    t = tile_grid(world_size)
    whole = np.array([t[i] * size[i] for i in range(3)], dtype=np.float64)
    centre = np.array(origin, dtype=np.float64) + 0.5 * np.array(size, dtype=np.float64)
    return tuple(((centre - 0.5 * whole) * voxel_width).tolist())


def aggregate(dist, seconds, voxels_per_rank, steps):
    """Whole-job throughput [Mvoxels/s] from the slowest rank's time (max-reduce)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        world = dist.get_world_size()
    else:
        world = 1
    t_max = float(t.item())
    return world * voxels_per_rank * steps / t_max / 1e6, t_max
