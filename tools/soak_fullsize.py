"""Full-size soak (GPU box): N map updates at BASELINE's 512^3 @ 0.05 m through libgie_hip.so and the
CPU oracle side by side, every update compared bit for bit (types, dist_sq, closest obstacle,
wave statistics).  Sensors alternate per --pattern so that the sparse (ray casting: tile lists,
direct pass Z) and dense (projective: volume sweeps, column pass Z, deep waves) forms follow each
other on the same map.  Usage: python tools/soak_fullsize.py --frames 40 --pattern rrp
(r = VLP-16 ray casting, p = VLP-16 projective)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd"), os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--pattern", default="rrp")
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--delta", type=int, default=8)
    ap.add_argument("--ray-sensor", default="vlp16", help="bench.SENSORS preset used for the ray-casting updates (vlp16 | lidar64)")
    ap.add_argument("--c5", action="store_true", help="BASELINE config 5's hash world under full observation instead of the lidar patterns: "
                                                      "every update seeds waves A, B and C (24 k / 36 k / 20 k visits at 512^3)")
    args = ap.parse_args()
    import bench
    import gie
    from gie import scenes
    from oracle_py import OracleMapper

    size = tuple(args.size)
    cfg = gie.make_config(0.05, size, cutoff_dist=2.0, fast_mode=False)
    fr = {} if args.c5 else {"r": bench.make_frames(scenes, 0.05, args.frames, 5, args.ray_sensor),
                             "p": bench.make_frames(scenes, 0.05, args.frames, 5, "vlp16_projective")}
    rings, az, phi_min, phi_inc, bins = bench.SENSORS["vlp16_projective"]
    kw = dict(theta_inc=2.0 * np.pi / bins, theta_min=-np.pi, phi_inc=np.radians(phi_inc), phi_min=np.radians(phi_min))
    a, b = OracleMapper(cfg), gie.Mapper(cfg)
    bad = 0
    t_cpu = t_gpu = 0.0
    import parity
    rng = np.random.default_rng(77)
    prev_pvt = None
    for k in range(args.frames):
        s = "c" if args.c5 else args.pattern[k % len(args.pattern)]
        if s == "c":
            pos, q = bench.c5_pose(scenes, k, 0.05)
            data = np.ascontiguousarray(scenes.hash_world_labels(scenes.local_pivot(pos, 0.05, size), size, k, seed=bench.C5["seed"], p_occ=bench.C5["p_occ"],
                                                                 toggle_frac=bench.C5["toggle_frac"]).astype(np.int8))
        else:
            pos, q, data, _ = fr[s][k]
        for m in (a, b):
            t0 = time.perf_counter()
            if s == "c":
                m.update(pos, q, "labels", data)
            elif s == "r":
                m.update(pos, q, "pointcloud", data)
            else:
                m.update(pos, q, "multiscan", data, **kw)
            if m is b:
                m.sync(); t_gpu += time.perf_counter() - t0
            else:
                t_cpu += time.perf_counter() - t0
        ra, rb = a.read_local(), b.read_local()
        sa, sb = a.stats(), b.stats()
        diff = [key for key in ("type", "dist_sq", "coc") if not np.array_equal(ra[key], rb[key])]
        if not np.allclose(ra["edt"], rb["edt"], rtol=1e-6, atol=0):
            diff.append("edt")
        diff += [key for key in ("seeds_a", "seeds_b", "seeds_c", "visits_a", "visits_b", "visits_c", "levels_a", "levels_b", "levels_c", "blocks_total")
                 if sa[key] != sb[key]]
        # the global map in and around the volume and in the slabs just left (round 6: a tskip tile's stored records live in the pair
        # plane until they are caught up or leave — gie_ops.h "deferred records")
        probes = [parity.probe_coords(a.pivot(), size, rng, n=20000, margin=12)]
        if prev_pvt is not None:
            probes.append(parity.probe_left_behind(prev_pvt, a.pivot(), size, rng))
        prev_pvt = a.pivot()
        for xyz in probes:
            if len(xyz):
                ga, gb = a.query_global(xyz), b.query_global(xyz)
                diff += ["global " + key for key in ("occ_val", "vox_type", "dist_sq", "coc") if not np.array_equal(ga[key], gb[key])]
        bad += bool(diff)
        print("update %3d %s known %.4f seeds %d/%d/%d visits %d/%d/%d levels_c %d blocks %d %s" % (
            k, s, float((rb["type"] != 0).mean()), sb["seeds_a"], sb["seeds_b"], sb["seeds_c"], sb["visits_a"], sb["visits_b"],
            sb["visits_c"], sb["levels_c"], sb["blocks_total"], "MISMATCH " + ",".join(diff) if diff else "ok"), flush=True)
        del ra, rb
    print("soak: %d updates, %d mismatching; oracle %.1f s, HIP %.3f s (incl. host upload + sync)" % (args.frames, bad, t_cpu, t_gpu))
    a.close(); b.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
