"""Spatial tiling with halo exchange (include/gie.h "spatial tiling across GPUs"): a 64x32x16
volume cut into two 32x32x16 tiles along x.  Both tiles see the SAME sensor frame (one robot);
after every map update they exchange one-voxel face layers and refine until no tile changes.

CPU: the device logic (test-only emulation) must equal the oracle's tiled restatement bit for
bit; the exchange must carry distance information across the cut; and the stitched tiles must be
(almost everywhere) the single-volume result.  GPU: the HIP library must equal the oracle the
same way (two mappers on the one GPU, host-side exchange)."""
import numpy as np
import pytest

import gie
from gie import scenes, tiling
from oracle_py import OracleMapper

TILE = (32, 32, 16)
WHOLE = (64, 32, 16)
W = 0.1
FR = 8


def _sensor_frames(n):
    world = scenes.BoxWorld(11, extent=(2.8, 1.4, 0.7), n_boxes=24, toggle_frac=0.3, ground_z=-0.6)
    out = []
    for k in range(n):
        pos, q = scenes.pose(k, W, delta_vox=1, yaw_deg=25.0)
        pts, _ = scenes.lidar_frame(world, k, pos, q, rings=32, az=720, phi_min_deg=-40.0, phi_inc_deg=2.5, max_range=10.0)
        img = scenes.range_image(pts, scan_num=360, ring_num=32, phi_min_deg=-40.0, phi_inc_deg=2.5)
        out.append((pos, q, img))
    return out


KW = dict(theta_inc=2.0 * np.pi / 360, theta_min=-np.pi, phi_inc=np.radians(2.5), phi_min=np.radians(-40.0))


def _run_tiled(make, exchange=True, device=None, fixed_rounds=0, sparse=False, sent=None, converged=0, stats=None):
    """converged = N: tiling.exchange_converged_local_device with at most N gated rounds (needs `device`; "cpu" for emulated
    mappers); `stats` (a list) then receives every mapper's round_stats() at the end."""
    cfg = gie.make_config(W, TILE, cutoff_dist=1.0)
    ms = [make(cfg), make(cfg)]
    for r, m in enumerate(ms):
        m.set_tile(tiling.tile_offset_voxels(r, 2, TILE), WHOLE)
    hist = []
    bufs = {}                                              # the face layers live as long as the mappers that read them
    try:
        for pos, q, img in _sensor_frames(FR):
            for m in ms:
                m.update(pos, q, "multiscan", img, tiled=True, **KW)
            if not exchange:
                for m in ms:
                    m.merge_end()
                rounds = 0
            elif converged:
                tiling.exchange_converged_local_device(ms, (2, 1, 1), device, max_rounds=converged, bufs=bufs)
                rounds = -1
            elif device is not None and fixed_rounds:
                tiling.exchange_rounds_local_device(ms, (2, 1, 1), device, rounds=fixed_rounds, bufs=bufs)
                rounds = -1
            elif device is not None:
                rounds = tiling.exchange_until_stable_local_device(ms, (2, 1, 1), device, bufs=bufs)
            else:
                rounds = tiling.exchange_until_stable_local(ms, (2, 1, 1), sparse=sparse, sent=sent)
            hist.append(([m.read_local() for m in ms], rounds, [m.pivot() for m in ms]))
        if stats is not None:
            stats.extend(m.round_stats() for m in ms)
    finally:
        for m in ms:
            m.close()
    return hist


def _run_whole(make):
    cfg = gie.make_config(W, WHOLE, cutoff_dist=1.0)
    m = make(cfg)
    out = []
    try:
        for pos, q, img in _sensor_frames(FR):
            m.update(pos, q, "multiscan", img, **KW)
            out.append((m.read_local(), m.pivot()))
    finally:
        m.close()
    return out


def _assert_same(ha, hb):
    assert len(ha) == len(hb)
    for k, ((ra, na, pa), (rb, nb, pb)) in enumerate(zip(ha, hb)):
        assert na == nb, "frame %d: %d vs %d refinement rounds" % (k, na, nb)
        assert pa == pb
        for t in range(2):
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(ra[t][key], rb[t][key]), "frame %d tile %d: %s differs" % (k, t, key)
            assert np.allclose(ra[t]["edt"], rb[t]["edt"], rtol=1e-6, atol=0)


def test_tiles_are_adjacent_and_share_the_sensor():
    cfg = gie.make_config(W, TILE)
    ms = [OracleMapper(cfg), OracleMapper(cfg)]
    for r, m in enumerate(ms):
        m.set_tile(tiling.tile_offset_voxels(r, 2, TILE), WHOLE)
        m.set_pose((0.3, -0.2, 0.1))
    whole = OracleMapper(gie.make_config(W, WHOLE))
    whole.set_pose((0.3, -0.2, 0.1))
    p0, p1, pw = ms[0].pivot(), ms[1].pivot(), whole.pivot()
    assert p0 == pw and p1 == (pw[0] + TILE[0], pw[1], pw[2])
    for m in ms + [whole]:
        m.close()


def test_emulated_tiled_matches_oracle_tiled(oracle_lib):
    from emu_py import EmuMapper
    _assert_same(_run_tiled(OracleMapper), _run_tiled(EmuMapper))


def _assert_converged_like(want, got, stats, max_rounds, tiles):
    """`got` ran gated rounds: the same maps as the oracle's "until no tile changes", and — the gate at work — exactly the
    oracle's rounds RAN in every tile (its count includes the last round, which seeds nothing), the rest of the `max_rounds`
    enqueued per update returned at once; no update left unconverged."""
    for k, ((ra, _, pa), (rb, _, pb)) in enumerate(zip(want, got)):
        assert pa == pb
        for t in range(tiles):
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(ra[t][key], rb[t][key]), (k, t, key)
    oracle_rounds = sum(n for _, n, _ in want)
    assert max(n for _, n, _ in want) <= max_rounds
    assert len(stats) == tiles
    for st in stats:
        assert st == {"rounds_enqueued": max_rounds * len(want), "rounds_run": oracle_rounds, "updates": len(want), "updates_unconverged": 0}, (st, oracle_rounds)


def test_emulated_tiled_gated_rounds_match_oracle_until_stable(oracle_lib):
    """SURVEY 8(e) "until no GPU changed" without the host: a fixed bound of rounds enqueued per update, each gated by the
    all-reduced "changed" word of the round before (gie_round_gate / gie_refine_dev / gie_round_end).  On the emulated device
    logic, words in host memory."""
    import torch
    from emu_py import EmuMapper
    stats = []
    want = _run_tiled(OracleMapper)
    got = _run_tiled(EmuMapper, device=torch.device("cpu"), converged=4, stats=stats)
    _assert_converged_like(want, got, stats, 4, 2)      # (this scene settles in the first refinement round; the hash world below needs 2-3)


def test_gated_rounds_report_an_update_the_bound_was_too_small_for(oracle_lib):
    import torch
    from emu_py import EmuMapper
    tile, frames = (24, 24, 24), 3
    want = _run_c5_tiled(OracleMapper, tile, frames)
    deep = sum(1 for _, n, _ in want if n > 1)             # updates whose first refinement round still seeded something somewhere
    assert deep > 0
    stats = []
    _run_c5_tiled(EmuMapper, tile, frames, device=torch.device("cpu"), converged=1, stats=stats)
    for st in stats:
        assert st["updates_unconverged"] == deep and st["rounds_run"] == len(want) == st["rounds_enqueued"]


def test_sparse_face_layers_are_interchangeable_with_dense_ones(oracle_lib):
    """gie_halo_export_sparse / gie_halo_import_sparse carry only the known voxels of a face layer.  Two lidar
    tiles exchanging sparse layers (the emulated device logic) against two oracle tiles exchanging dense ones: the same maps,
    the same number of refinement rounds, fewer bytes."""
    from emu_py import EmuMapper
    dense, sparse = [], []
    ref = _run_tiled(OracleMapper, sent=dense)
    _assert_same(ref, _run_tiled(EmuMapper, sparse=True, sent=sparse))
    # (24 bytes per known voxel against 20 per voxel: the shared face of this scene is 62 % known — a face seen by a lidar from afar is a few per cent)
    assert len(dense) == len(sparse) and 0 < sum(sparse) < 0.9 * sum(dense), (sum(sparse), sum(dense))


def test_exchange_carries_information_and_approaches_the_single_volume(oracle_lib):
    with_x = _run_tiled(OracleMapper, exchange=True)
    without = _run_tiled(OracleMapper, exchange=False)
    whole = _run_whole(OracleMapper)
    lowered = 0
    bad_with = bad_without = total = 0
    for (ra, na, _), (rb, _, _), (rw, _) in zip(with_x, without, whole):
        stitched_t = np.concatenate([ra[0]["type"], ra[1]["type"]], axis=2)
        assert np.array_equal(stitched_t != 0, rw["type"] != 0)             # the same voxels are known
        for t in range(2):
            known = (ra[t]["type"] != 0) & (rb[t]["type"] != 0)
            assert (ra[t]["dist_sq"][known] <= rb[t]["dist_sq"][known]).all()   # exchange only lowers
            lowered += int((ra[t]["dist_sq"][known] < rb[t]["dist_sq"][known]).sum())
        sw = np.concatenate([ra[0]["dist_sq"], ra[1]["dist_sq"]], axis=2)
        so = np.concatenate([rb[0]["dist_sq"], rb[1]["dist_sq"]], axis=2)
        k = (rw["type"] != 0) & (stitched_t != 0) & (rw["dist_sq"] < 900000)
        total += int(k.sum())
        bad_with += int((sw[k] != rw["dist_sq"][k]).sum())
        bad_without += int((so[k] != rw["dist_sq"][k]).sum())
    assert lowered > 0
    assert bad_with < bad_without            # the exchange moves the tiles towards the single-volume field
    # measured: 30 of 113 000 voxel-frames differ (by <= 0.5 voxel: BFS propagation is not an exact EDT)
    assert bad_with <= 0.002 * total, (bad_with, bad_without, total)


@pytest.mark.gpu
def test_hip_tiled_matches_oracle_tiled(oracle_lib):
    _assert_same(_run_tiled(OracleMapper), _run_tiled(gie.Mapper))


@pytest.mark.gpu
def test_hip_sparse_face_layers(oracle_lib):
    dense, sparse = [], []
    ref = _run_tiled(OracleMapper, sent=dense)
    _assert_same(ref, _run_tiled(gie.Mapper, sparse=True, sent=sparse))
    assert 0 < sum(sparse) < 0.9 * sum(dense), (sum(sparse), sum(dense))


@pytest.mark.gpu
def test_hip_tiled_device_resident_exchange(oracle_lib):
    """the *_dev halo entry points (what the RCCL path uses): layers never leave the GPU"""
    import torch
    _assert_same(_run_tiled(OracleMapper), _run_tiled(gie.Mapper, device=torch.device("cuda", 0)))


@pytest.mark.gpu
def test_hip_tiled_gated_rounds(oracle_lib):
    """The exchange `bench.py --gpus N` runs by default (tiling.exchange_converged_*): rounds gated on the device."""
    import torch
    stats = []
    want = _run_tiled(OracleMapper)
    got = _run_tiled(gie.Mapper, device=torch.device("cuda", 0), converged=4, stats=stats)
    _assert_converged_like(want, got, stats, 4, 2)


@pytest.mark.gpu
def test_hip_tiled_stream_ordered_rounds(oracle_lib):
    """The exchange the multi-GPU bench uses: a fixed number of rounds enqueued on the mappers' own
    streams (gie_get_stream, gie_refine without a seed count), the host never waits.  Rounds
    beyond convergence change nothing, so four rounds must equal "until stable" of the oracle."""
    import torch
    want = _run_tiled(OracleMapper)
    assert max(n for _, n, _ in want) <= 4
    got = _run_tiled(gie.Mapper, device=torch.device("cuda", 0), fixed_rounds=4)
    for k, ((ra, _, pa), (rb, _, pb)) in enumerate(zip(want, got)):
        assert pa == pb
        for t in range(2):
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(ra[t][key], rb[t][key]), (k, t, key)


@pytest.mark.gpu
def test_eight_mappers_share_one_device(oracle_lib):
    """Eight tiles as eight mappers (eight streams) on ONE GPU: their waves kernels need all their
    workgroups resident at once, so launches from different mappers must not overlap (they are
    chained through a per-device event).  Before that, this ran into the grid-barrier time-out."""
    import torch
    size = (64, 64, 64)
    grid = tiling.tile_grid(8)
    whole = tuple(grid[i] * size[i] for i in range(3))
    cfg = gie.make_config(0.1, size, cutoff_dist=1.0)
    world = scenes.BoxWorld(3, extent=(5.0, 5.0, 5.0), n_boxes=60, toggle_frac=0.3)
    def run(make, device):
        ms = []
        for r in range(8):
            m = make(cfg); m.set_tile(tiling.tile_offset_voxels(r, 8, size), whole); ms.append(m)
        out = None
        bufs = {}
        try:
            for k in range(4):
                pos, q = scenes.pose(k, 0.1, delta_vox=1, yaw_deg=10.0)
                pts, _ = scenes.lidar_frame(world, k, pos, q, rings=32, az=360, phi_min_deg=-40.0, phi_inc_deg=2.5, max_range=10.0)
                for m in ms:
                    m.update(pos, q, "pointcloud", pts, tiled=True)
                if device is None:
                    tiling.exchange_until_stable_local(ms, grid)
                else:
                    tiling.exchange_rounds_local_device(ms, grid, device, rounds=6, bufs=bufs)
            out = [m.read_local() for m in ms]
        finally:
            for m in ms:
                m.close()
        return out
    want = run(OracleMapper, None)
    got = run(gie.Mapper, torch.device("cuda", 0))
    for t in range(8):
        for key in ("type", "dist_sq", "coc"):
            assert np.array_equal(want[t][key], got[t][key]), (t, key)


def _run_c5_tiled(make, tile, frames, device=None, fixed_rounds=0, delta_vox=8, converged=0, stats=None, world=8):
    """BASELINE config 5 in small: the sensor-less hash world (full observation, a quarter of the obstacles toggles
    per frame) on 2x2x2 tiles (world = 8; 2: two tiles along x), every tile fed the label plane of its own part of the world."""
    grid = tiling.tile_grid(world)
    whole = tuple(grid[i] * tile[i] for i in range(3))
    cfg = gie.make_config(0.05, tile, cutoff_dist=0.5)
    ms = []
    for r in range(world):
        m = make(cfg); m.set_tile(tiling.tile_offset_voxels(r, world, tile), whole); ms.append(m)
    out, bufs = [], {}
    try:
        for k in range(frames):
            pos, q = scenes.pose(k, 0.05, delta_vox=delta_vox, yaw_deg=2.0)
            for r, m in enumerate(ms):
                pvt = scenes.local_pivot(pos, 0.05, tile, tiling.tile_offset_voxels(r, world, tile))
                m.set_pose(pos, q)
                assert tuple(m.pivot()) == tuple(pvt)
                m.ogm_labels(scenes.hash_world_labels(pvt, tile, k, seed=5, p_occ=0.01).astype(np.int8))
                m.step_begin_tiled()
            if converged:
                tiling.exchange_converged_local_device(ms, grid, device, max_rounds=converged, bufs=bufs)
                rounds = -1
            elif device is None:
                rounds = tiling.exchange_until_stable_local(ms, grid)
            elif fixed_rounds:
                tiling.exchange_rounds_local_device(ms, grid, device, rounds=fixed_rounds, bufs=bufs)
                rounds = -1
            else:
                rounds = tiling.exchange_until_stable_local_device(ms, grid, device, bufs=bufs)
            out.append(([m.read_local() for m in ms], rounds, [tuple(m.pivot()) for m in ms]))
        if stats is not None:
            stats.extend(m.round_stats() for m in ms)
    finally:
        for m in ms:
            m.close()
    return out


def _run_c5_whole(make, tile, frames, delta_vox=8):
    whole = tuple(2 * t for t in tile)
    m = make(gie.make_config(0.05, whole, cutoff_dist=0.5))
    out = []
    try:
        for k in range(frames):
            pos, q = scenes.pose(k, 0.05, delta_vox=delta_vox, yaw_deg=2.0)
            m.set_pose(pos, q)
            m.ogm_labels(scenes.hash_world_labels(scenes.local_pivot(pos, 0.05, whole), whole, k, seed=5, p_occ=0.01).astype(np.int8))
            m.step()
            out.append((m.read_local(), tuple(m.pivot())))
    finally:
        m.close()
    return out


def _stitch(tiles, key):
    """tiles: rank r = tx + 2 (ty + 2 tz) -> one array [Z][Y][X] of the 2x2x2 arrangement"""
    return np.concatenate([np.concatenate([np.concatenate([tiles[tx + 2 * (ty + 2 * tz)][key] for tx in range(2)], axis=2)
                                           for ty in range(2)], axis=1) for tz in range(2)], axis=0)


def _c5_tiled_vs_whole(tiled, whole):
    """What the exchange promises: the same voxels known, and a stitched field that is the single volume's almost
    everywhere (BFS propagation across a cut is not an exact EDT: a few voxels end half a voxel above)."""
    diff = total = 0
    for (rt, _, pt), (rw, pw) in zip(tiled, whole):
        assert pt[0] == pw
        assert np.array_equal(_stitch(rt, "type") != 0, rw["type"] != 0)
        a, b = _stitch(rt, "dist_sq"), rw["dist_sq"]
        known = rw["type"] != 0
        total += int(known.sum())
        diff += int((a[known] != b[known]).sum())
        assert (a[known] >= b[known]).mean() > 0.999      # the tiles never know more than the single volume
    assert diff <= 0.01 * total, (diff, total)
    return diff, total


def test_c5_hash_world_tiled_emulation(oracle_lib):
    """2x2x2 tiles of 24^3 (CPU: oracle against the emulated device logic, and against the single 48^3 volume).  Six updates: from the
    third on the tiles' inner 8x8x8 tiles leave their records to the pair plane (deferred records), and the face layers that
    cross them are exported through gie_deferred_coc."""
    from emu_py import EmuMapper
    tile, frames = (24, 24, 24), 6
    want = _run_c5_tiled(OracleMapper, tile, frames)
    got = _run_c5_tiled(EmuMapper, tile, frames)
    for k, ((ra, na, pa), (rb, nb, pb)) in enumerate(zip(want, got)):
        assert na == nb and pa == pb
        for t in range(8):
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(ra[t][key], rb[t][key]), (k, t, key)
    _c5_tiled_vs_whole(want, _run_c5_whole(OracleMapper, tile, frames))
    # ... and with the gated rounds bench.py enqueues per update (its own bound)
    import torch
    import bench
    stats = []
    gated = _run_c5_tiled(EmuMapper, tile, frames, device=torch.device("cpu"), converged=bench.HALO_MAX_ROUNDS, stats=stats)
    _assert_converged_like(want, gated, stats, bench.HALO_MAX_ROUNDS, 8)
    assert any(n > 1 for _, n, _ in want) and any(n < bench.HALO_MAX_ROUNDS for _, n, _ in want)      # rounds ran beyond the first AND were cut short


@pytest.mark.gpu
def test_c5_hash_world_2x2x2_gated_rounds_with_the_bench_bound(oracle_lib):
    """the exchange the multi-GPU bench times by default — at most bench.HALO_MAX_ROUNDS refinement rounds per
    update, gated on the device by the all-reduced "changed" word — against the tiled oracle's "until no tile changes" on
    BASELINE config 5's arrangement (2x2x2 tiles of 64^3): the same maps bit for bit, the oracle's number of rounds ran, and no
    update was left unconverged."""
    import torch
    import bench
    tile, frames = (64, 64, 64), 4
    stats = []
    want = _run_c5_tiled(OracleMapper, tile, frames)
    got = _run_c5_tiled(gie.Mapper, tile, frames, device=torch.device("cuda", 0), converged=bench.HALO_MAX_ROUNDS, stats=stats)
    _assert_converged_like(want, got, stats, bench.HALO_MAX_ROUNDS, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["until_stable", "stream_ordered"])
def test_c5_hash_world_2x2x2_tiles_of_64_cubed_on_one_gpu(oracle_lib, form):
    """BASELINE config 5's arrangement at 1/8 scale per axis: 2x2x2 tiles of 64^3 (one 128^3 volume), hash world, full
    observation, 25 % toggling, all eight mappers on one GPU with the device-resident exchange — until no tile
    changes, and as the fixed stream-ordered rounds the multi-GPU bench enqueues.  Against the tiled oracle bit for
    bit, and against the single 128^3 volume."""
    import torch
    tile, frames = (64, 64, 64), 4
    want = _run_c5_tiled(OracleMapper, tile, frames)
    assert max(n for _, n, _ in want) <= 6
    got = _run_c5_tiled(gie.Mapper, tile, frames, device=torch.device("cuda", 0), fixed_rounds=6 if form == "stream_ordered" else 0)
    for k, ((ra, na, pa), (rb, nb, pb)) in enumerate(zip(want, got)):
        assert pa == pb and (form == "stream_ordered" or na == nb)
        for t in range(8):
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(ra[t][key], rb[t][key]), (k, t, key)
            assert np.allclose(ra[t]["edt"], rb[t]["edt"], rtol=1e-6, atol=0)
    whole = _run_c5_whole(gie.Mapper, tile, frames)
    diff, total = _c5_tiled_vs_whole(got, whole)
    print("tiled vs single volume: %d of %d voxel-frames differ" % (diff, total))


class _InProcessTransport:
    """Stands in for torch.distributed in test_rank_exchange_code_path: the P2P calls that
    tiling.exchange_rounds_device makes (P2POp / isend / irecv / batch_isend_irecv, work.wait())
    between two tiles that live in two THREADS of this process on one GPU.  A receive is a
    device-to-device copy on the receiver's current stream, ordered after an event on the sender's
    stream; a send's wait() orders the sender's stream after that copy (as RCCL would: the send
    buffer is free again)."""

    def __init__(self):
        import threading
        self.cv = threading.Condition()
        self.posted = {}      # (src, dst, seq) -> (tensor, event on the sender's stream)
        self.copied = {}      # (src, dst, seq) -> event on the receiver's stream
        self.reduce_in = {}   # all-reduce round -> {rank: value}
        self.meet = threading.Barrier(2)

    def view(self, rank):
        return _RankView(self, rank)


class _Work:
    def __init__(self, fn):
        self.fn = fn

    def wait(self):
        self.fn()


class _RankView:
    isend, irecv = "isend", "irecv"

    class ReduceOp:
        SUM = "sum"
        MAX = "max"

    def all_reduce(self, t, op=None, group=None):
        """sum / max of a one-element tensor over the two ranks (host-synchronous — RCCL's would not be; the data flow is the same)"""
        n = self.seq[("ar",)] = self.seq.get(("ar",), 0) + 1
        with self.s.cv:
            self.s.reduce_in.setdefault(n, {})[self.rank] = int(t.item())
        self.s.meet.wait(timeout=60)
        t.fill_((max if op == "max" else sum)(self.s.reduce_in[n].values()))

    def __init__(self, shared, rank):
        self.s, self.rank, self.seq = shared, rank, {}

    class P2POp:
        def __init__(self, op, tensor, peer, group=None):
            self.op, self.tensor, self.peer = op, tensor, peer

    def batch_isend_irecv(self, ops):
        import torch
        s, works = self.s, []
        cur = torch.cuda.current_stream()
        for o in ops:
            key = (self.rank, o.peer, o.op)
            n = self.seq[key] = self.seq.get(key, 0) + 1
            if o.op == self.isend:
                ev = torch.cuda.Event(); ev.record(cur)
                with s.cv:
                    s.posted[(self.rank, o.peer, n)] = (o.tensor, ev); s.cv.notify_all()

                def wait_send(k=(self.rank, o.peer, n), cur=cur):
                    with s.cv:
                        assert s.cv.wait_for(lambda: k in s.copied, timeout=60)
                    cur.wait_event(s.copied[k])
                works.append(_Work(wait_send))
            else:
                k = (o.peer, self.rank, n)
                with s.cv:
                    assert s.cv.wait_for(lambda: k in s.posted, timeout=60)
                src, ev = s.posted[k]
                cur.wait_event(ev)
                o.tensor.copy_(src, non_blocking=True)
                done = torch.cuda.Event(); done.record(cur)
                with s.cv:
                    s.copied[k] = done; s.cv.notify_all()
                works.append(_Work(lambda: None))
        return works


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["stream_ordered", "until_stable", "until_stable_sparse", "converged"])
def test_rank_exchange_code_path(oracle_lib, form):
    """tiling.exchange_rounds_device — the function bench.py calls once per map update on every
    rank of a multi-GPU run — with an in-process transport instead of RCCL: two tiles, two
    threads, one GPU.  Same export / transfer / import / refine sequence on the mappers' own
    streams (torch.cuda.ExternalStream, cached P2P op list and pointer tables); the result must
    equal the in-process stream-ordered rounds, i.e. the oracle's (test above).  "until_stable" =
    tiling.exchange_until_stable_device, the host-synchronised form bench.py falls back to; "_sparse" = the same with sparse face
    layers (counts first, then the known voxels only: gie_halo_export_sparse_dev / gie_halo_import_sparse_dev)."""
    import threading
    import torch
    device = torch.device("cuda", 0)
    want = _run_tiled(gie.Mapper, device=device, fixed_rounds=4 if form in ("stream_ordered", "converged") else 0)
    cfg = gie.make_config(W, TILE, cutoff_dist=1.0)
    frames = _sensor_frames(FR)
    shared = _InProcessTransport()
    step_barrier = threading.Barrier(2)
    hist, errors = [[], []], []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            m = gie.Mapper(cfg)
            m.set_tile(tiling.tile_offset_voxels(rank, 2, TILE), WHOLE)
            dist, bufs = shared.view(rank), {}
            try:
                for pos, q, img in frames:
                    m.update(pos, q, "multiscan", img, tiled=True, **KW)
                    if form == "stream_ordered":
                        tiling.exchange_rounds_device(m, dist, rank, 2, device, bufs, rounds=4)
                    elif form == "converged":        # bench.py's default for N > 1
                        tiling.exchange_converged_device(m, dist, rank, 2, device, bufs, max_rounds=4)
                    else:                            # bench.py's fall-back: host-synchronised rounds until no tile changes
                        tiling.exchange_until_stable_device(m, dist, rank, 2, device, bufs, sparse=form.endswith("sparse"))
                    hist[rank].append((m.read_local(), m.pivot()))
                    step_barrier.wait(timeout=120)
                if form == "converged":
                    st = m.round_stats()
                    assert st["updates_unconverged"] == 0 and st["rounds_enqueued"] == 4 * len(frames) and FR <= st["rounds_run"] < 4 * len(frames), st
            finally:
                m.close()
        except Exception as e:                       # noqa: BLE001 — reported by the main thread
            errors.append((rank, repr(e)))
            step_barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    assert len(hist[0]) == FR and len(hist[1]) == FR
    for k, (ra, _, pa) in enumerate(want):
        for t in range(2):
            assert pa[t] == hist[t][k][1]
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(ra[t][key], hist[t][k][0][key]), (k, t, key)
