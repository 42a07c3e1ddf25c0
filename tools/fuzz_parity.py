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


def random_scenario(rng, i):
    import parity
    dims = [8, 16, 24, 29, 32, 37, 40, 48, 56, 64, 72, 96]
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
    if sensor in ("depth", "mixed", "multiscan", "lidar_points", "scan2d"):
        ext = max(size) * voxel
        kw["extent"] = (0.6 * ext + 1.0, 0.6 * ext + 1.0, 0.4 * size[2] * voxel + 0.5)
    return parity.Scenario("fuzz%d" % i, size, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--only", type=int, default=-1, help="run scenario number N of the seed only")
    args = ap.parse_args()
    import gie
    import parity
    from oracle_py import OracleMapper
    if args.emu:
        from emu_py import EmuMapper as Under
    else:
        Under = gie.Mapper
    rng = np.random.default_rng(args.seed)
    t0, i, ok = time.time(), 0, 0
    while time.time() - t0 < 60.0 * args.minutes:
        sc = random_scenario(rng, i)
        i += 1
        if args.only >= 0 and i - 1 != args.only:
            continue
        desc = "%s %s voxel %.2f %s frames %d delta %d cutoff %.1f fast %d planner %d boxes %d retain %d turn %d" % (
            sc.name, sc.size, sc.voxel, sc.sensor, sc.frames, sc.delta_vox, sc.cutoff_dist, sc.fast_mode, sc.for_motion_planner, sc.ext_boxes, sc.retain, sc.turn)
        try:
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
