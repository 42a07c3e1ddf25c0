"""world_size-2 CPU test (gloo) of the rank plumbing bench.py uses for N > 1: tile assignment,
per-rank independent mappers, barrier + max-reduce timing, whole-job aggregation."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import time
    import torch.distributed as dist
    import gie
    from gie import scenes, tiling
    from emu_py import EmuMapper
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tile = (32, 32, 16)
    origin, size = tiling.tile_of_rank(rank, world, tile)
    off = tiling.tile_centre_offset(rank, world, tile, 0.1)
    cfg = gie.make_config(0.1, size, cutoff_dist=1.0)
    m = EmuMapper(cfg)
    world_model = scenes.BoxWorld(3, extent=(3, 3, 1), n_boxes=20)
    dist.barrier()
    t0 = time.perf_counter()
    steps = 3
    for k in range(steps):
        pos, q = scenes.pose(k, 0.1, delta_vox=2, yaw_deg=20.0)
        depth = scenes.depth_frame(world_model, k, pos, q, rows=60, cols=80, fx=65, fy=65, cx=39.5, cy=29.5)
        # every rank sees the shared sensor pose; its local volume is centred on its own tile
        m.update((pos[0] + off[0], pos[1] + off[1], pos[2] + off[2]), q, "depth", depth, cx=39.5, cy=29.5, fx=65, fy=65,
                 valid_nan=True)
    dist.barrier()
    dt = time.perf_counter() - t0 + 0.01 * rank          # rank 1 is "slower"
    value, t_max = tiling.aggregate(dist, dt, size[0] * size[1] * size[2], steps)
    pv = m.pivot()
    m.close()
    out.put((rank, origin, value, t_max, dt, pv))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    from gie import tiling
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, o0, v0, t0, d0, p0), (r1, o1, v1, t1, d1, p1) = res
    assert o0 == (0, 0, 0) and o1 == (32, 0, 0)               # 2 ranks: split along x
    assert v0 == pytest.approx(v1) and t0 == pytest.approx(t1)  # everyone agrees on the aggregate
    assert t0 == pytest.approx(max(d0, d1))                    # max over ranks
    assert v0 == pytest.approx(2 * 32 * 32 * 16 * 3 / t0 / 1e6)
    assert p1[0] - p0[0] == 32 and p1[1] == p0[1] and p1[2] == p0[2]   # adjacent, non-overlapping local volumes


def _worker_halo(rank, world, port, out, sparse=False):
    sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import gie
    from gie import tiling
    from emu_py import EmuMapper
    import test_tiling_halo as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = gie.make_config(T.W, T.TILE, cutoff_dist=1.0)
    m = EmuMapper(cfg)
    m.set_tile(tiling.tile_offset_voxels(rank, world, T.TILE), T.WHOLE)
    res = []
    for pos, q, img in T._sensor_frames(4):
        m.update(pos, q, "multiscan", img, tiled=True, **T.KW)
        rounds = tiling.exchange_until_stable(m, dist, rank, world, sparse=sparse)
        r = m.read_local()
        res.append((rounds, r["type"].copy(), r["dist_sq"].copy(), r["coc"].copy()))
    m.close()
    out.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.parametrize("sparse", [False, True], ids=["dense_layers", "sparse_layers"])
def test_two_ranks_halo_exchange_matches_in_process_exchange(oracle_lib, sparse):
    """The torch.distributed exchange (isend/irecv + all-reduce of the seed count) must give each
    rank exactly what the in-process exchange of the oracle gives the corresponding tile — with dense face layers and
    with sparse ones (entry counts first, then the known voxels only)."""
    import test_tiling_halo as T
    from oracle_py import OracleMapper
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_halo, args=(r, 2, port, q, sparse)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = T._run_tiled(OracleMapper)[:4]
    for k, (reads, rounds, _) in enumerate(ref):
        for t in range(2):
            g_rounds, g_type, g_dist, g_coc = got[t][k]
            assert g_rounds == rounds
            assert np.array_equal(g_type, reads[t]["type"])
            assert np.array_equal(g_dist, reads[t]["dist_sq"])
            assert np.array_equal(g_coc, reads[t]["coc"])


def _worker_gated(rank, world, port, out, max_rounds):
    """One rank of the exchange `bench.py --gpus N` runs by default (tiling.exchange_converged_device), on the emulated device logic
    over gloo: the "device words" are host tensors, the all-reduce(max) of the "changed" word is gloo's."""
    sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import gie
    from gie import scenes, tiling
    from emu_py import EmuMapper
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tile = (24, 24, 24)
    grid = tiling.tile_grid(world)
    m = EmuMapper(gie.make_config(0.05, tile, cutoff_dist=0.5))
    off = tiling.tile_offset_voxels(rank, world, tile)
    m.set_tile(off, tuple(grid[i] * tile[i] for i in range(3)))
    res, bufs = [], {}
    for k in range(3):
        pos, q = scenes.pose(k, 0.05, delta_vox=8, yaw_deg=2.0)
        m.set_pose(pos, q)
        m.ogm_labels(scenes.hash_world_labels(scenes.local_pivot(pos, 0.05, tile, off), tile, k, seed=5, p_occ=0.01).astype(np.int8))
        m.step_begin_tiled()
        tiling.exchange_converged_device(m, dist, rank, world, torch.device("cpu"), bufs, max_rounds=max_rounds)
        r = m.read_local()
        res.append((r["type"].copy(), r["dist_sq"].copy(), r["coc"].copy()))
    st = m.round_stats()
    m.close()
    out.put((rank, res, st))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_gated_rounds_match_the_oracle_until_stable(oracle_lib, world):
    """VERDICT r4 #3 on the CPU: two / four ranks (four: 2x2x1 tiles, information crosses two cuts and needs 2-3 rounds) run the gated rounds of the multi-GPU bench (export / send-receive / import / refine /
    all-reduce(max) of the changed word, at most bench.HALO_MAX_ROUNDS per update) on the hash world; each rank must hold what the
    oracle's "until no tile changes" gives its tile, and exactly the oracle's rounds must have RUN (the others hit a closed gate)."""
    import bench
    import test_tiling_halo as T
    from oracle_py import OracleMapper
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_gated, args=(r, world, port, q, bench.HALO_MAX_ROUNDS)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: (res, st) for r, res, st in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = T._run_c5_tiled(OracleMapper, (24, 24, 24), 3, world=world)
    oracle_rounds = sum(n for _, n, _ in ref)
    assert world == 2 or (1 < max(n for _, n, _ in ref) <= bench.HALO_MAX_ROUNDS and min(n for _, n, _ in ref) < bench.HALO_MAX_ROUNDS)      # rounds ran beyond the first, and were cut short
    for t in range(world):
        res, st = got[t]
        assert st == {"rounds_enqueued": 3 * bench.HALO_MAX_ROUNDS, "rounds_run": oracle_rounds, "updates": 3, "updates_unconverged": 0}, st
        for k, (reads, _, _) in enumerate(ref):
            assert np.array_equal(res[k][0], reads[t]["type"])
            assert np.array_equal(res[k][1], reads[t]["dist_sq"])
            assert np.array_equal(res[k][2], reads[t]["coc"])


def _worker_transport(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))
    import torch
    import torch.distributed as dist
    from gie import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    # no GPU here: the RCCL pre-flight fails on every rank, and every rank must come back with the same answer
    info = tiling.init_transport(torch, dist, rank, world, torch.device("cuda", 0), want="nccl", preflight_timeout_s=20)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # the control plane works
    out.put((rank, info["backend"], info["group"] is None, info["note"], float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_preflight_falls_back_on_every_rank_alike():
    """bench.py --gpus N: the face layers go over RCCL only if a pre-flight (group creation, an all-reduce, a ring send / receive)
    passes on EVERY rank; otherwise all ranks stage them through the host over gloo and say why (the first
    execution of the RCCL path must not be able to kill the driver's run)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_transport, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, backend, no_group, note, mx in got:
        assert backend == "gloo" and no_group and mx == 1.0
        assert note and "RCCL pre-flight failed" in note


def test_tile_layouts():
    from gie import tiling
    assert tiling.tile_grid(1) == (1, 1, 1) and tiling.tile_grid(2) == (2, 1, 1)
    assert tiling.tile_grid(4) == (2, 2, 1) and tiling.tile_grid(8) == (2, 2, 2)
    seen = set()
    for r in range(8):
        o, s = tiling.tile_of_rank(r, 8, (512, 512, 512))
        assert all(v % 8 == 0 for v in o)
        seen.add(o)
    assert len(seen) == 8 and max(max(o) for o in seen) == 512   # 2x2x2 tiles of 512^3 = 1024^3
    offs = np.array([tiling.tile_centre_offset(r, 8, (512, 512, 512), 0.05) for r in range(8)])
    assert np.allclose(offs.mean(axis=0), 0.0) and np.allclose(np.abs(offs), 12.8)
    with pytest.raises(ValueError):
        tiling.tile_grid(3)
