"""CPU suite: the device-side per-voxel / per-frontier-entry logic (gie_ops.h) and the frame
orchestration (gie_api.inc.h), run through the test-only sequential emulation, must equal the
oracle bit for bit on every stage of multi-frame scenarios that exercise waves A, B and C."""
import numpy as np
import pytest

import parity
from emu_py import EmuMapper
from oracle_py import OracleMapper

SCENARIOS = [
    parity.Scenario("depth", (48, 40, 24), sensor="depth", frames=14, delta_vox=5, yaw_deg=47.0),
    parity.Scenario("raycast", (40, 40, 20), sensor="pointcloud", frames=12, delta_vox=5, yaw_deg=47.0),
    parity.Scenario("vlp16", (48, 48, 16), sensor="multiscan", frames=12, delta_vox=5, yaw_deg=10.0),
    # a volume tall enough for most of it to lie above / below the lidar's field of view: the box test of the multiscan OGM (round 6)
    parity.Scenario("vlp16_tall", (64, 56, 72), voxel=0.05, sensor="multiscan", frames=5, delta_vox=4, yaw_deg=25.0, extent=(2.5, 2.5, 2.0)),
    parity.Scenario("scan2d", (40, 40, 8), sensor="scan2d", frames=6, delta_vox=3, yaw_deg=10.0),
    parity.Scenario("mixed", (48, 40, 24), sensor="mixed", frames=12, delta_vox=5, yaw_deg=47.0),
    parity.Scenario("fast_mode", (48, 40, 24), sensor="mixed", frames=12, delta_vox=5, yaw_deg=47.0, fast_mode=True),
    parity.Scenario("planner_boxes", (40, 40, 24), sensor="mixed", frames=9, delta_vox=4, yaw_deg=30.0,
                    for_motion_planner=True, ext_boxes=True, min_h=-0.8, max_h=1.0),
    parity.Scenario("cutoff_small", (32, 32, 16), sensor="lidar_points", frames=12, delta_vox=6, yaw_deg=5.0,
                    cutoff_dist=0.5),
    parity.Scenario("odd_dims", (37, 29, 11), sensor="mixed", frames=9, delta_vox=3, yaw_deg=33.0),
    parity.Scenario("flat_2d", (40, 40, 1), sensor="scan2d", frames=5, delta_vox=3, yaw_deg=10.0),
    # ray casting in thin volumes: rays cut by the maximum length 0.707*X*w (X = 16), and rays that leave the volume long before they end
    parity.Scenario("thin_x", (16, 200, 8), sensor="lidar_points", frames=6, delta_vox=5, yaw_deg=40.0, lidar_az=720),
    parity.Scenario("thin_y", (200, 16, 8), sensor="lidar_points", frames=6, delta_vox=5, yaw_deg=40.0, lidar_az=720),
    # a volume taller than 512 voxels: more than 64 z tiles per tile column (the upper tiles were never listed nor flagged)
    parity.Scenario("tall_z", (16, 24, 600), voxel=0.05, sensor="labels", frames=4, delta_vox=5, yaw_deg=2.0, seed=8, cutoff_dist=1.0, p_occ=0.004),
    # BASELINE C3's wave parameters (ugv yaml: cutoff 100 m => no cutoff at all, full waves A+B), small volume
    parity.Scenario("c3_no_cutoff", (56, 56, 20), voxel=0.1, sensor="multiscan", frames=10, delta_vox=6, yaw_deg=12.0,
                    cutoff_dist=100.0, extent=(5.0, 5.0, 1.5)),
    # BASELINE C5's sensor-less world (hash occupancy, full observation, 25 % toggling, block-aligned motion) and a
    # denser / misaligned variant; the pre-classified scan goes through gie_ogm_labels
    parity.Scenario("c5_hash_world", (48, 48, 32), voxel=0.05, sensor="labels", frames=8, delta_vox=8, yaw_deg=2.0, seed=5,
                    cutoff_dist=2.0, p_occ=0.01),
    parity.Scenario("c5_dense_odd", (40, 36, 20), voxel=0.05, sensor="labels", frames=8, delta_vox=3, yaw_deg=2.0, seed=6,
                    cutoff_dist=0.5, p_occ=0.05, toggle=0.5),
    # block-pool lifecycle (gie_config.retain_radius_blocks): blocks more than R blocks behind the volume are erased and their
    # slots recycled.  A drive out and back over the same ground (what was erased comes back as unknown space), a lidar drive
    # whose waves A / B walk through remembered space right up to the erased zone, and R = 1 on a misaligned volume; the probes
    # reach well past the retention zone
    parity.Scenario("retain_c5_out_and_back", (32, 32, 16), voxel=0.05, sensor="labels", frames=26, delta_vox=8, yaw_deg=2.0, seed=5,
                    cutoff_dist=1.0, p_occ=0.01, retain=2, turn=9, probe_margin=60),
    # obstacle-free scans in between, on the move: known voxels that are not committed (EMPTY pairs) while their stored pair is
    # still owed from an earlier update of the same stay in the volume (gie_commit_pair / gie_pair_flush_voxel)
    parity.Scenario("blink_empty_scans", (32, 28, 12), voxel=0.05, sensor="labels_blink", frames=18, delta_vox=5, yaw_deg=2.0, seed=9,
                    cutoff_dist=1.0, p_occ=0.004, toggle=0.3, probe_margin=30),
    parity.Scenario("retain_lidar", (48, 48, 16), sensor="multiscan", frames=16, delta_vox=7, yaw_deg=10.0, retain=1, probe_margin=40),
    parity.Scenario("retain_odd_r1", (37, 29, 11), sensor="mixed", frames=12, delta_vox=5, yaw_deg=33.0, retain=1, turn=5, probe_margin=40),
    # a robot that turns round and meets the blocks it has erased again, in a volume whose block table reaches a cell beyond the
    # retention box (round-4 fuzz on the GPU, seed 51 #30).  The device's allocation asks the table of the fuse before first; the
    # emulation does not, but checks for every cell that the table names what the hash finds (gie_cell_prev_slot, gie_emu.cpp)
    parity.Scenario("retain_turn_back", (72, 64, 96), voxel=0.1, sensor="mixed", frames=10, delta_vox=8, yaw_deg=28.425771268056188, seed=870,
                    cutoff_dist=0.5, extent=(6.76, 6.76, 4.34), toggle=0.25, lidar_az=180, p_occ=0.003, retain=1, turn=3),
]


@pytest.mark.parametrize("sc", SCENARIOS, ids=[s.name for s in SCENARIOS])
def test_emulated_device_logic_matches_oracle(oracle_lib, sc):
    stats = parity.run_and_compare(sc, OracleMapper, EmuMapper)
    assert len(stats) == sc.frames


@pytest.mark.parametrize("sc", parity.UNEVEN_DRIVES, ids=[s.name for s in parity.UNEVEN_DRIVES])
@pytest.mark.parametrize("production", [False, True], ids=["staged", "production"])
def test_uneven_drives_catch_up_the_deferred_records_emulation(oracle_lib, sc, production):
    """a new tile that straddles the old volume's face must have its deferred records caught up too."""
    parity.run_and_compare(sc, OracleMapper, EmuMapper, production=production)


@pytest.mark.parametrize("sc", [s for s in SCENARIOS if s.name in ("raycast", "mixed", "c5_hash_world")], ids=lambda s: s.name)
def test_production_sequence_emulation(oracle_lib, sc):
    """set_pose / ogm / step() only (no readers between the stages)."""
    parity.run_and_compare(sc, OracleMapper, EmuMapper, production=True)


def test_waves_are_exercised(oracle_lib):
    """The scenarios above are only meaningful if all three wavefronts actually run."""
    sc = parity.Scenario("vlp16", (48, 48, 16), sensor="multiscan", frames=12, delta_vox=5, yaw_deg=10.0)
    stats = parity.run_and_compare(sc, OracleMapper, EmuMapper)
    assert sum(s["visits_a"] for s in stats) > 0
    assert sum(s["visits_b"] for s in stats) > 0
    assert sum(s["visits_c"] for s in stats) > 0


def test_long_drive_on_a_fixed_pool_emulation(oracle_lib):
    """2 000 map updates of the c5 world in a straight line on a pool that holds the retention zone and nothing more: without
    recycling the drive needs ~18 000 blocks, the pool has 400.  The map equals the oracle's wherever it is probed, the live
    block count is bounded and equal on both sides, and the hash table survives its periodic rebuilds (every 64 updates)."""
    import numpy as np
    import gie as _gie
    from gie import scenes
    size, voxel, R = (16, 16, 16), 0.05, 2
    cfg = _gie.make_config(voxel, size, cutoff_dist=0.5, fast_mode=False, retain_radius_blocks=R, max_blocks=400)
    a, b = OracleMapper(cfg), EmuMapper(cfg)
    rng = np.random.default_rng(11)
    peak = 0
    for i in range(2000):
        pos, q = scenes.pose(i, voxel, delta_vox=8, yaw_deg=1.0)
        lab = scenes.hash_world_labels(scenes.local_pivot(pos, voxel, size), size, i, seed=5, p_occ=0.02, toggle_frac=0.25).astype(np.int8)
        for m in (a, b):
            m.update(pos, q, "labels", lab)
        if i % 97 == 0 or i == 1999:
            xyz = parity.probe_coords(a.pivot(), size, rng, n=3000, margin=48)
            ga, gb = a.query_global(xyz), b.query_global(xyz)
            for key in ("occ_val", "vox_type", "dist_sq", "coc"):
                assert np.array_equal(ga[key], gb[key]), (i, key)
            ra, rb = a.read_local(), b.read_local()
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(ra[key], rb[key]), (i, key)
            sa, sb = a.stats(), b.stats()
            assert sa["blocks_total"] == sb["blocks_total"]
            peak = max(peak, sb["blocks_total"])
    assert peak <= (3 + 2 * R + 1) ** 3          # the retention zone: (2 + 1 blocks of volume +-1 voxel, + R on either side)^3
    a.close(); b.close()


def test_one_merge_per_map_update_emulation():
    """gie_merge without a gie_fuse since the last merge is a call-order error (the seed counters and the wavefront kernel's
    barrier words belong to the frame clear), and it leaves the mapper usable."""
    import numpy as np
    import gie as _gie
    sc = parity.Scenario("twice", (24, 24, 8), sensor="labels", voxel=0.05, frames=2, delta_vox=4, p_occ=0.02)
    m = EmuMapper(sc.config())
    frames = list(sc.frames_iter())
    pos, q, kind, data, kw = frames[0]
    m.set_pose(pos, q); m.ogm_labels(data); m.fuse(); m.batch_edt(); m.merge()
    with pytest.raises(RuntimeError) as e:
        m.merge()
    assert "no gie_fuse since the last merge" in str(e.value)
    with pytest.raises(RuntimeError):
        m.fuse()                                          # and no scan, no fuse
    pos, q, kind, data, kw = frames[1]
    m.update(pos, q, kind, data)
    m.sync()
    m.close()


def test_block_pool_overflow_fails_loudly_emulation():
    """A pool that is too small is a sticky, reported error (never a silent wrong map)."""
    import gie as _gie
    from emu_py import EmuMapper
    from parity import Scenario
    sc = Scenario("tiny_pool", (40, 40, 16), sensor="depth", frames=2)
    cfg = _gie.make_config(sc.voxel, sc.size, cutoff_dist=1.0, max_blocks=5)
    m = EmuMapper(cfg)
    with pytest.raises(RuntimeError) as e:
        for pos, q, kind, data, kw in sc.frames_iter():
            m.update(pos, q, kind, data, **kw)
            m.sync()
    assert "block pool" in str(e.value)
    m.close()


def test_edge_inputs_emulation(oracle_lib):
    """Empty scans, NaN / Inf / far / duplicate / zero-length points, negative far-away poses,
    images without a valid reading: the emulated device logic equals the oracle on all of them."""
    import edge_inputs
    edge_inputs.run(OracleMapper, EmuMapper)


def test_hash_world_numpy_equals_torch():
    """The C5 label generator is written once and run on numpy (tests) or torch (bench.py, on the GPU)."""
    import numpy as np
    import torch
    from gie import scenes
    for pvt, frame in (((-37, 12, -5), 0), ((100000, -70001, 33), 7)):
        a = scenes.hash_world_labels(pvt, (24, 20, 12), frame, seed=5)
        b = scenes.hash_world_labels(pvt, (24, 20, 12), frame, seed=5,
                                     arange=lambda n: torch.arange(n, dtype=torch.int64), where=lambda c, x, y: torch.where(c, x, y))
        assert np.array_equal(np.asarray(a), b.numpy())
    lab = scenes.hash_world_labels((0, 0, 0), (128, 128, 64), 0, seed=5, p_occ=0.01)
    frac = float((lab == 2).mean())
    assert 0.006 < frac < 0.011                   # p_occ minus the toggled-off share
    nxt = scenes.hash_world_labels((0, 0, 0), (128, 128, 64), 1, seed=5, p_occ=0.01)
    changed = float(((lab == 2) != (nxt == 2)).sum()) / float(((lab == 2) | (nxt == 2)).sum())
    assert 0.15 < changed < 0.35                  # about a quarter of the obstacles toggle per frame


def test_rejected_pose_leaves_the_mapper_untouched_and_fuse_needs_a_scan():
    """gie_set_pose validates before it changes anything (a rejected pose does not bump the frame counter or the
    pivots); gie_fuse without a scan since the last fuse is an error, not a re-fuse of stale labels."""
    import numpy as np
    import gie as _gie
    cfg = _gie.make_config(0.1, (24, 24, 8), cutoff_dist=1.0)
    m = EmuMapper(cfg)
    try:
        m.set_pose((1.0, 2.0, 0.5))
        pvt, frame = m.pivot(), m.stats()["frame"]
        for bad in ((1.0e9, 0.0, 0.0), (0.0, float("nan"), 0.0), (0.0, 0.0, -95000.0)):
            with pytest.raises(RuntimeError):
                m.set_pose(bad)
            assert m.pivot() == pvt and m.stats()["frame"] == frame
        with pytest.raises(RuntimeError) as e:
            m.fuse()
        assert "no scan" in str(e.value)
        m.ogm_labels(np.ones((8, 24, 24), np.int8))
        m.fuse()
        with pytest.raises(RuntimeError):
            m.fuse()                                  # the scan was consumed
    finally:
        m.close()


def test_random_scenarios_emulation_matches_oracle(oracle_lib):
    """tools/fuzz_parity.py's generator (random volume shapes, sensors, drives, cut-offs, modes, block retention) on the emulated device
    logic: a fixed seed's first scenarios (the GPU form ran 4 185 of them: profiles/r03_fuzz_parity_seed11_summary.txt)."""
    import importlib.util
    import numpy as np
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from emu_py import EmuMapper
    rng = np.random.default_rng(7)
    ran = 0
    for i in range(40):
        sc = fz.random_scenario(rng, i)
        if sc.size[0] * sc.size[1] * sc.size[2] > 64 * 64 * 40:
            continue                                   # (keeps the CPU suite short; the generator's stream stays the same)
        parity.run_and_compare(sc, OracleMapper, EmuMapper, production=bool(i % 2))
        ran += 1
    assert ran >= 15


IRREGULAR = [
    # (scenario name, fuse-only frames, frames with the changed-block flags on)
    ("mixed", (3, 6), ()), ("blink_empty_scans", (3, 6, 7, 11), ()), ("c5_hash_world", (2, 3, 5), ()),
    ("retain_odd_r1", (2, 5, 6), ()),
    ("mixed", (), (3, 4, 8)), ("blink_empty_scans", (), (2, 3, 4, 9, 10, 14)), ("blink_empty_scans", (5, 12), (3, 4, 9, 10, 13)),
]


@pytest.mark.parametrize("name,fuse_only,stream_on", IRREGULAR, ids=["%s-f%s-s%s" % (n, "_".join(map(str, f)), "_".join(map(str, s))) for n, f, s in IRREGULAR])
def test_irregular_call_orders_emulation(oracle_lib, name, fuse_only, stream_on):
    """gie_fuse without a merge behind it, and runs that change between the fused sweep and the reference's order: the stored
    pairs a fused update leaves out must still reach the map (they did not when a fuse was not followed by a merge)."""
    sc = [s for s in SCENARIOS if s.name == name][0]
    parity.run_irregular(sc, OracleMapper, EmuMapper, fuse_only=set(fuse_only), stream_on=set(stream_on))


@pytest.mark.parametrize("fuse_only,stream_on", [((3,), ()), ((3,), (5,)), ((2, 3), ())], ids=["f3", "f3-s5", "f2_3"])
def test_fuse_without_merge_then_jump_emulation(oracle_lib, fuse_only, stream_on):
    parity.run_irregular(parity.FUSE_ONLY_THEN_JUMP, OracleMapper, EmuMapper, fuse_only=set(fuse_only), stream_on=set(stream_on))


def _jumping_robot(make_b, updates=150):
    """A robot that lands on new ground in every update (jumps of 100 voxels) with block retention on: every update erases all it
    held.  The tombstones of the erased blocks use up the hash table's EMPTY cells — 125 per update against a table of 512
    cells here — long before the 64-update rebuild period is over; lookups must stay finite and the map right."""
    import numpy as np
    import gie as _gie
    from gie import scenes
    size, voxel = (16, 16, 16), 0.05
    cfg = _gie.make_config(voxel, size, cutoff_dist=0.5, retain_radius_blocks=1, max_blocks=120)
    a, b = OracleMapper(cfg), make_b(cfg)
    rng = np.random.default_rng(5)
    try:
        for i in range(updates):
            pos, q = scenes.pose(i, voxel, delta_vox=100, yaw_deg=1.0)
            lab = scenes.hash_world_labels(scenes.local_pivot(pos, voxel, size), size, i, seed=5, p_occ=0.02, toggle_frac=0.25).astype(np.int8)
            for m in (a, b):
                m.update(pos, q, "labels", lab)
            if i % 13 == 0 or i == updates - 1:
                xyz = parity.probe_coords(a.pivot(), size, rng, n=2000, margin=24)
                ga, gb = a.query_global(xyz), b.query_global(xyz)
                for key in ("occ_val", "vox_type", "dist_sq", "coc"):
                    assert np.array_equal(ga[key], gb[key]), (i, key)
                assert a.stats()["blocks_total"] == b.stats()["blocks_total"]
    finally:
        a.close(); b.close()


def test_jumping_robot_keeps_the_hash_table_alive_emulation(oracle_lib):
    _jumping_robot(EmuMapper)


def test_byte_parallel_occupancy_filter():
    """gie_fuse_row8_labels (the block-row fuse kernel's byte-parallel form of set_hashvoxel_occ_val for labelled scans) against the
    per-voxel filter: every (label 0..3, stored type 0..3, stored occupancy 0..255) in every byte position, thresholds 127..255,
    and random rows."""
    import ctypes as C
    import numpy as np
    import emu_py
    emu_py.load()
    lib = C.CDLL(emu_py.EMU_SO)
    lib.gie_emu_fuse_row8.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.gie_emu_fuse_voxel.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]

    def one(thresh, label, occ, ty):
        o, t = C.c_int(), C.c_int()
        lib.gie_emu_fuse_voxel(thresh, label, occ, ty, C.byref(o), C.byref(t))
        return o.value, t.value
    def row(thresh, labels, occs, tys):
        pack = lambda v: sum((int(b) & 0xff) << (8 * i) for i, b in enumerate(v))
        no, ny = C.c_uint64(), C.c_uint64()
        lib.gie_emu_fuse_row8(thresh, pack(labels), pack(occs), pack(tys), C.byref(no), C.byref(ny))
        return [(no.value >> (8 * i)) & 0xff for i in range(8)], [(ny.value >> (8 * i)) & 0xff for i in range(8)]
    rng = np.random.default_rng(3)
    for thresh in (127, 180, 200, 254, 255):
        table = {(l, t, o): one(thresh, l, o, t) for l in range(4) for t in range(4) for o in range(256)}
        # every combination in byte 0..7, the other bytes random
        for pos in range(8):
            for l in range(4):
                for t in range(4):
                    for o in range(0, 256, 1 if pos in (0, 7) else 17):
                        labels, occs, tys = rng.integers(0, 4, 8), rng.integers(0, 256, 8), rng.integers(0, 4, 8)
                        labels[pos], occs[pos], tys[pos] = l, o, t
                        no, ny = row(thresh, labels, occs, tys)
                        for i in range(8):
                            assert (no[i], ny[i]) == table[(int(labels[i]), int(tys[i]), int(occs[i]))], (thresh, pos, i, labels, occs, tys, no, ny)


WAVE_C_MODEL = ["depth", "mixed", "c3_no_cutoff", "c5_hash_world", "c5_dense_odd", "retain_lidar", "retain_turn_back", "odd_dims", "planner_boxes"]


@pytest.mark.parametrize("halo", [1, 2], ids=["halo_as_of_round_start", "halo_live"])
@pytest.mark.parametrize("name", WAVE_C_MODEL)
def test_device_schedule_of_wave_c_matches_oracle(oracle_lib, name, halo):
    """The emulation's SECOND statement of wave C: the tile rounds as the device runs them (seeds assigned inside round 0, halos as of
    the start of the round, proposals that do not beat the pair in sight dropped, proposals across tile borders merged a round
    later) against the oracle's sequential schedule — every stage, every frame, and the wave statistics."""
    import emu_py
    sc = [s for s in SCENARIOS if s.name == name][0]
    emu_py.wave_c_model(halo)          # 1: a tile's halo is the plane as of the start of the round; 2: the live plane — the two ends of what a halo read can see on the device
    try:
        parity.run_and_compare(sc, OracleMapper, EmuMapper)
    finally:
        emu_py.wave_c_model(False)


def test_device_schedule_of_wave_c_needs_the_unfiltered_round_0(oracle_lib):
    """What the retain-focused fuzz of round 4 found on the GPU (seed 83 #63), on the CPU: with the halo filter also applied
    across tile borders in round 0 — the kernel before the fix — the device schedule leaves ONE voxel at the distance of the seed
    wave B put on it (182) where the oracle reaches 179 through the neighbour across the tile border; without it they agree."""
    import emu_py
    sc = parity.Scenario("wave_c_seed_above_stale_pair", (96, 120, 160), voxel=0.2, sensor="mixed", frames=9, delta_vox=10, yaw_deg=30.029303880177924,
                         seed=742, cutoff_dist=100.0, extent=(20.2, 20.2, 13.3), toggle=0.5, lidar_az=360, p_occ=0.01, retain=2, turn=5, probe_margin=40)
    emu_py.wave_c_model(True, r0filter=True)
    try:
        with pytest.raises(AssertionError, match="frame 8: post-merge dist_sq differs in 1 voxels"):
            parity.run_and_compare(sc, OracleMapper, EmuMapper)
        emu_py.wave_c_model(True, r0filter=False)
        parity.run_and_compare(sc, OracleMapper, EmuMapper)
    finally:
        emu_py.wave_c_model(False)


@pytest.mark.parametrize("retain", [0, 2])
def test_halo_import_between_pose_and_fuse_keeps_the_block_table_honest(oracle_lib, retain):
    """the ABI only asks gie_halo_import* for a pose, so an import may run between gie_set_pose and gie_fuse.  Its
    ghost-block pass rebuilds the block table at the NEW pose's origin; the next gie_fuse took that table for "the table of the fuse
    before" at the OLD origin and the shortcut of k_cell_alloc named slots of other blocks (with retain == 0 nothing checked them).
    The emulation's allocation asserts, for every cell, that what the previous table names is what the hash finds — it aborted on
    this sequence before the fix.  And the ghosts land on voxels of the previous volume whose stored pairs were still owed (fused
    Mark + commit): those are flushed first now.  Two adjacent tiles that exchange their x faces right after the pose, then run a
    plain update; the emulated device logic against the oracle."""
    import gie
    from gie import scenes, tiling
    tile = (32, 24, 16)
    whole = (64, 24, 16)
    cfg = gie.make_config(0.05, tile, cutoff_dist=0.5, fast_mode=False, retain_radius_blocks=retain)
    probes = np.stack(np.meshgrid(np.arange(-40, 110, 3), np.arange(-14, 14, 3), np.arange(-10, 10, 3), indexing="ij"), -1).reshape(-1, 3).astype(np.int32)
    out = []
    for make in (OracleMapper, EmuMapper):
        ms = [make(cfg), make(cfg)]
        try:
            offs = [tiling.tile_offset_voxels(r, 2, tile) for r in range(2)]
            for r, m in enumerate(ms):
                m.set_tile(offs[r], whole)
            for k, shift in enumerate((0, 16, 19, 40, 33)):
                pos = (np.float32(shift * 0.05), np.float32(0.0), np.float32(0.0))
                for m in ms:
                    m.set_pose(pos)
                if k > 0:                                   # the shared face, exchanged BEFORE this pose's fuse (the maps' state of the update before)
                    a, b = ms[0].halo_export(1), ms[1].halo_export(0)
                    ms[1].halo_import(0, a)
                    ms[0].halo_import(1, b)
                for r, m in enumerate(ms):
                    m.ogm_labels(scenes.hash_world_labels(scenes.local_pivot(pos, 0.05, tile, offs[r]), tile, k, seed=7, p_occ=0.03).astype(np.int8))
                    m.fuse(); m.batch_edt(); m.merge()
            out.append([(m.read_local(), m.query_global(probes), m.stats()["blocks_total"]) for m in ms])
        finally:
            for m in ms:
                m.close()
    for (ra, ga, ba), (rb, gb, bb) in zip(*out):
        assert ba == bb
        for key in ("type", "dist_sq", "coc"):
            assert np.array_equal(ra[key], rb[key]), key
        assert np.array_equal(ga, gb), int((ga != gb).sum())


def test_device_reader_entry_points_on_the_emulated_backend(oracle_lib):
    """gie_query_global_dev / gie_read_costmap_dev / gie_costmap_publish + _acquire through the shared orchestration code
    (gie_api.inc.h) on the emulated backend, where a "device pointer" is host memory: the bytes of the host forms."""
    sc = parity.Scenario("dev_readers_cpu", (40, 32, 16), sensor="mixed", frames=3, delta_vox=5, yaw_deg=20.0)
    cfg = sc.config()
    a, b = OracleMapper(cfg), EmuMapper(cfg)
    X, Y, Z = sc.size
    try:
        for pos, q, kind, data, kw in sc.frames_iter():
            for m in (a, b):
                m.update(pos, q, kind, data, **kw)
            b.costmap_publish()
            pv = np.array(b.pivot())
            probes = np.ascontiguousarray((np.stack(np.meshgrid(np.arange(-9, X + 9, 4), np.arange(-9, Y + 9, 4), np.arange(-9, Z + 9, 4), indexing="ij"), -1).reshape(-1, 3) + pv).astype(np.int32))
            want = a.query_global(probes)
            out = np.zeros(probes.shape[0], want.dtype)
            b.query_global_dev(probes.ctypes.data, probes.shape[0], out.ctypes.data)
            assert out.tobytes() == want.tobytes()
            pa, _ = a.read_costmap()
            cm = np.zeros(X * Y * Z * 8, np.uint8)
            b.read_costmap_dev(cm.ctypes.data)
            assert cm.tobytes() == pa.tobytes()
            assert b.costmap_acquire().tobytes() == pa.tobytes()
    finally:
        a.close(); b.close()
