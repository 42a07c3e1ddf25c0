"""Randomised parity hunt (GPU box): seeded random scenarios — volume shape, voxel width, sensor mix, drive, cut-off, fast mode,
planner boxes, block retention — through the oracle and libgie_hip.so with every stage compared (tests/parity.py), for a
number of minutes.  Prints one line per scenario; exits non-zero on the first mismatch, naming the seed.
    python tools/fuzz_parity.py --minutes 10 [--seed 1] [--emu]     (--emu: the test-only host emulation instead of the HIP library)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd"), os.path.join(ROOT, "tests")]


def random_scenario(rng, i, big=False, focus=None, drive=None):
    import parity
    dims = [96, 120, 125, 128, 152, 157, 160, 192] if big else [8, 13, 16, 21, 24, 29, 32, 35, 37, 40, 43, 48, 51, 56, 64, 72, 96]
    size = tuple(int(rng.choice(dims)) for _ in range(3))
    if rng.random() < 0.15:
        size = (size[0], size[1], int(rng.choice([1, 8, 11, 16])))
    sensor = str(rng.choice(["depth", "pointcloud" if False else "mixed", "multiscan", "lidar_points", "labels", "labels", "labels_blink", "scan2d"]))
    voxel = float(rng.choice([0.05, 0.1, 0.1, 0.2]))
    kw = dict(voxel=voxel, sensor=sensor, frames=int(rng.integers(3, 11)), delta_vox=int(rng.integers(0, 10)), yaw_deg=float(rng.uniform(0, 60)),
              seed=int(rng.integers(1, 1000)), cutoff_dist=float(rng.choice([0.5, 1.0, 2.0, 100.0])) * (voxel / 0.1 if voxel < 0.1 else 1.0),
              fast_mode=bool(rng.random() < 0.15), for_motion_planner=bool(rng.random() < 0.15), ext_boxes=bool(rng.random() < 0.1),
              p_occ=float(rng.choice([0.003, 0.01, 0.03])), toggle=float(rng.choice([0.0, 0.25, 0.5])),
              retain=int(rng.choice([0, 0, 0, 1, 2])), turn=int(rng.choice([0, 0, 3, 5])), probe_margin=int(rng.choice([12, 40])),
              lidar_az=int(rng.choice([180, 360, 720])))
    if drive == "random" or (drive is None and rng.random() < 0.5):
        # a drive that changes speed and direction (round 6): mostly steps of 0-1 voxels on every axis, now and then a jump
        # of 6-9 either way on some axes -- the tskip / prev_shift / catch-up state machine sees slow updates followed by a jump
        steps = []
        for _ in range(kw["frames"]):
            st = rng.integers(-1, 2, size=3)
            if rng.random() < 0.3:
                jump = rng.integers(6, 10, size=3) * rng.choice([-1, 1], size=3)
                st = np.where(rng.random(3) < 0.6, jump, st)
            steps.append(tuple(int(v) for v in st))
        kw["steps"] = steps
    if focus == "retain":          # block erasure on a robot that turns round: long drives, small radii, frequent turns
        kw.update(retain=int(rng.choice([1, 1, 2, 3])), turn=int(rng.choice([2, 3, 4, 5, 7])), frames=int(rng.integers(8, 17)),
                  delta_vox=int(rng.integers(3, 17)), probe_margin=40)
    if sensor in ("depth", "mixed", "multiscan", "lidar_points", "scan2d"):
        ext = max(size) * voxel
        kw["extent"] = (0.6 * ext + 1.0, 0.6 * ext + 1.0, 0.4 * size[2] * voxel + 0.5)
    return parity.Scenario("fuzz%d" % i, size, **kw)


def run_tiled(rng, i, Under, OracleMapper):
    """A random TILED hash world: 2 / 4 / 8 block-aligned tiles of one volume, face layers exchanged in process until no tile
    changes (dense or sparse layers), the oracle's tiles against the tiles of the mapper under test, bit for bit."""
    import gie
    from gie import scenes, tiling
    world = int(rng.choice([2, 4, 8]))
    grid = tiling.tile_grid(world)
    tile = tuple(int(rng.choice([8, 16, 24, 32, 40])) for _ in range(3))
    whole = tuple(grid[a] * tile[a] for a in range(3))
    voxel = float(rng.choice([0.05, 0.1]))
    cut = float(rng.choice([0.3, 0.5, 1.0, 100.0])) * (voxel / 0.05)
    frames, delta, seed = int(rng.integers(3, 8)), int(rng.integers(0, 10)), int(rng.integers(1, 1000))
    p_occ, toggle, sparse = float(rng.choice([0.003, 0.01, 0.03])), float(rng.choice([0.0, 0.25, 0.5])), bool(rng.random() < 0.5)
    desc = "tiled%d %d tiles of %s voxel %.2f frames %d delta %d cutoff %.1f p_occ %.3f toggle %.2f %s" % (
        i, world, tile, voxel, frames, delta, cut, p_occ, toggle, "sparse layers" if sparse else "dense layers")
    cfg = gie.make_config(voxel, tile, cutoff_dist=cut)

    def run(make, sp):
        ms, out = [], []
        for r in range(world):
            m = make(cfg); m.set_tile(tiling.tile_offset_voxels(r, world, tile), whole); ms.append(m)
        try:
            for k in range(frames):
                pos, q = scenes.pose(k, voxel, delta_vox=delta, yaw_deg=2.0)
                for r, m in enumerate(ms):
                    pvt = scenes.local_pivot(pos, voxel, tile, tiling.tile_offset_voxels(r, world, tile))
                    m.set_pose(pos, q)
                    m.ogm_labels(scenes.hash_world_labels(pvt, tile, k, seed=seed, p_occ=p_occ, toggle_frac=toggle).astype(np.int8))
                    m.fuse(); m.batch_edt(); m.merge_begin_tiled()
                rounds = tiling.exchange_until_stable_local(ms, grid, sparse=sp)
                out.append((rounds, [m.read_local() for m in ms], [m.stats() for m in ms]))
        finally:
            for m in ms:
                m.close()
        return out
    want, got = run(OracleMapper, False), run(Under, sparse)
    for k, ((ra, la, sa), (rb, lb, sb)) in enumerate(zip(want, got)):
        assert ra == rb, "%s frame %d: %d vs %d refinement rounds" % (desc, k, ra, rb)
        for t in range(world):
            for key in ("type", "dist_sq", "coc"):
                assert np.array_equal(la[t][key], lb[t][key]), "%s frame %d tile %d: %s differs" % (desc, k, t, key)
            for key in ("visits_a", "visits_b", "blocks_total"):       # (wave C runs once per refinement round: its counters are per launch)
                assert sa[t][key] == sb[t][key], "%s frame %d tile %d: stat %s %d != %d" % (desc, k, t, key, sa[t][key], sb[t][key])
    return desc, [sum(s[t][key] for _, _, s in want for t in range(world)) for key in ("visits_a", "visits_b", "visits_c")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiled", action="store_true", help="random tiled hash worlds (2 / 4 / 8 tiles, in-process exchange) instead of single volumes")
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--big", action="store_true", help="volume sides of 96 ... 192 voxels (thousands of active blocks per wave round)")
    ap.add_argument("--only", type=int, default=-1, help="run scenario number N of the seed only")
    ap.add_argument("--device-wave-c", action="store_true", help="with --emu: the emulation runs its model of the DEVICE's wave C tile rounds (gie_emu.cpp be_wave_c_device) instead of the canonical statement")
    ap.add_argument("--irregular", action="store_true", help="every third scenario through parity.run_irregular: random updates stop after gie_fuse, random updates run with gie_stream_enable")
    ap.add_argument("--drive", default=None, choices=[None, "line", "random"], help="line: constant steps along x (rounds 1-5); random: per-frame steps on all axes with jumps; default: half and half")
    ap.add_argument("--focus", default=None, choices=[None, "retain"], help="retain: every scenario erases blocks (retain_radius_blocks 1-3) on a drive that turns round")
    args = ap.parse_args()
    import gie
    import parity
    from oracle_py import OracleMapper
    if args.emu:
        from emu_py import EmuMapper as Under
        if args.device_wave_c:
            import emu_py
            emu_py.wave_c_model(2 if os.environ.get("FUZZ_HALO_LIVE") else 1)
    else:
        Under = gie.Mapper
    rng = np.random.default_rng(args.seed)
    t0, i, ok = time.time(), 0, 0
    while args.tiled and time.time() - t0 < 60.0 * args.minutes:
        try:
            desc, v = run_tiled(rng, i, Under, OracleMapper)
        except AssertionError as e:
            print("MISMATCH seed %d #%d: %s" % (args.seed, i, str(e).splitlines()[0]), flush=True)
            sys.exit(1)
        i += 1; ok += 1
        print("ok %s | visits %d/%d/%d" % (desc, v[0], v[1], v[2]), flush=True)
    while not args.tiled and time.time() - t0 < 60.0 * args.minutes:
        sc = random_scenario(rng, i, args.big, args.focus, args.drive)
        i += 1
        if args.only >= 0 and i - 1 != args.only:
            continue
        desc = "%s %s voxel %.2f %s frames %d delta %d cutoff %.1f fast %d planner %d boxes %d retain %d turn %d" % (
            sc.name, sc.size, sc.voxel, sc.sensor, sc.frames, sc.delta_vox, sc.cutoff_dist, sc.fast_mode, sc.for_motion_planner, sc.ext_boxes, sc.retain, sc.turn) + (" steps %s" % (sc.steps,) if sc.steps else "")
        try:
            if args.irregular and i % 3 == 0:
                # the staged ABI off the beaten path: some updates stop after gie_fuse, some run with the changed-block flags on
                fo = set(int(k) for k in range(1, sc.frames) if rng.random() < 0.25)
                so = set(int(k) for k in range(1, sc.frames) if rng.random() < 0.2)
                desc += " fuse-only %s stream %s" % (sorted(fo), sorted(so))
                parity.run_irregular(sc, OracleMapper, Under, fuse_only=fo, stream_on=so)
                st = []
            else:
                st = parity.run_and_compare(sc, OracleMapper, Under, production=bool(i % 2))
        except AssertionError as e:
            print("MISMATCH seed %d #%d: %s\n   %s" % (args.seed, i - 1, desc, str(e).splitlines()[0]), flush=True)
            sys.exit(1)
        ok += 1
        va = sum(s["visits_a"] for s in st); vb = sum(s["visits_b"] for s in st); vc = sum(s["visits_c"] for s in st)
        print("ok %s | visits %d/%d/%d" % (desc, va, vb, vc), flush=True)
        if args.only >= 0:
            break
    print("fuzz: %d scenarios of seed %d, 0 mismatching, %.1f min" % (ok, args.seed, (time.time() - t0) / 60.0))


if __name__ == "__main__":
    main()
