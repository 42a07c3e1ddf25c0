"""Pre-flight of bench.py's plan (CPU): the block pool a bench run is given must hold every block the run can reach,
for every (steps, warmup) a driver may pass and however many regions the box's speed makes the run time.

Round 2's driver bench died in exactly this gap: the c5 robot drove in a straight line forever, the default pool held
about 157 updates, `--steps 20 --warmup 5` needed 165, and only `--steps 10 --warmup 3` had ever been run."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from gie import scenes, tiling  # noqa: E402

COMMANDS = [(10, 3), (20, 5), (50, 10), (100, 20), (1, 0)]


def blocks_reached(size, voxel, updates, tile_off=(0, 0, 0), ghost=0):
    """Bounding box, in 8x8x8 blocks, of everything the c5 volume (+ `ghost` voxels around it) overlaps in `updates` frames."""
    lo = np.full(3, 1 << 30)
    hi = np.full(3, -(1 << 30))
    for i in range(updates):
        pos, _ = bench.c5_pose(scenes, i, voxel)
        pvt = np.array(scenes.local_pivot(pos, voxel, size, tile_off))
        lo = np.minimum(lo, (pvt - ghost) >> 3)
        hi = np.maximum(hi, (pvt + np.array(size) - 1 + ghost) >> 3)
    return int(np.prod(hi - lo + 1))


@pytest.mark.parametrize("steps,warmup", COMMANDS)
@pytest.mark.parametrize("world", [1, 8])
def test_pool_holds_every_frame_the_command_can_reach(steps, warmup, world):
    size, voxel = (512, 512, 512), 0.05
    # the slowest plausible box times ONE region (>= 0.5 s), the fastest all MAX_REGIONS of them: plan for the latter
    updates = bench.planned_updates(warmup, steps)
    assert updates == warmup + steps * bench.MAX_REGIONS + steps
    pool = bench.pool_blocks("c5", size, updates)
    assert 0 < pool <= 4000000
    for rank in range(world):
        off = tiling.tile_offset_voxels(rank, world, size) if world > 1 else (0, 0, 0)
        need = blocks_reached(size, voxel, updates, off, ghost=1 if world > 1 else 0)
        assert need <= pool, "rank %d: %d blocks reached, pool %d" % (rank, need, pool)


def test_c5_drive_is_bounded():
    """Out and back: frame 2 * C5_TURN stands where frame 0 stood, and no frame is farther out than C5_TURN steps."""
    xs = [float(bench.c5_pose(scenes, i, 0.05)[0][0]) for i in range(5 * bench.C5_TURN)]
    assert xs[0] == xs[2 * bench.C5_TURN] == xs[4 * bench.C5_TURN] == 0.0
    assert max(xs) == pytest.approx(bench.C5_TURN * bench.C5["delta_vox"] * 0.05)
    assert min(xs) == 0.0
    # consecutive frames are exactly delta_vox voxels apart (the volume moves every frame: waves are seeded every step)
    d = np.abs(np.diff(np.array(xs))) / 0.05
    assert np.allclose(d, bench.C5["delta_vox"], atol=1e-3)


def test_small_volume_runs_the_whole_plan_on_the_planned_pool(oracle_lib):
    """The driver's command (--steps 20 --warmup 5) at its worst case of 12 regions, replayed at 32^3 through the sequential
    emulation of the device logic on a pool sized by the same rule: no capacity error, and the map holds exactly the blocks the plan counted."""
    import gie
    from emu_py import EmuMapper
    size, voxel = (32, 32, 32), 0.05
    updates = bench.planned_updates(5, 20)
    pool = bench.pool_blocks("c5", size, updates) - 4096      # without the fixed margin: the rule itself has to hold
    cfg = gie.make_config(voxel, size, cutoff_dist=0.5, fast_mode=False, max_blocks=pool)
    m = EmuMapper(cfg)
    for i in range(updates):
        pos, q = bench.c5_pose(scenes, i, voxel)
        lab = scenes.hash_world_labels(scenes.local_pivot(pos, voxel, size), size, i, seed=5, p_occ=0.01, toggle_frac=0.25)
        m.update(pos, q, "labels", lab.astype(np.int8))
        if i % 16 == 0:
            m.sync()                      # raises on a capacity error
    m.sync()
    used = m.stats()["blocks_total"]
    assert used <= pool
    assert used == blocks_reached(size, voxel, updates)       # full observation: every block the volume ever overlapped exists
    m.close()


def test_failure_still_prints_one_json_line(tmp_path):
    """Without a GPU bench.py cannot run: it must still print exactly one JSON line carrying an "error" key."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"], capture_output=True, text=True,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""), timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert p.returncode != 0
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert "error" in j and j["value"] is None
