"""Cost of the changed-block stream (gie_stream_changed) next to the map update itself, on the
bench workload.  Off the headline metric (the reference turns streaming off for speed too,
README.md:154): prints one JSON line per run."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gie-mapping_amd")]

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--voxel", type=float, default=0.05)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--sensor", default="vlp16")
    args = ap.parse_args()
    import torch
    import bench
    import gie
    from gie import scenes
    size = tuple(args.size)
    rings, az, phi_min, phi_inc, bins = bench.SENSORS[args.sensor]
    frames = bench.make_frames(scenes, args.voxel, args.frames, 5, args.sensor)
    dev = torch.device("cuda", 0)
    d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
    out = {}
    for track in (0, 1):
        m = gie.Mapper(gie.make_config(args.voxel, size, cutoff_dist=2.0, fast_mode=False))
        m.stream_enable(bool(track))
        upd, strm, nblk = [], [], []
        for i, (pos, q, pts, _) in enumerate(frames):
            m.sync()
            t0 = time.perf_counter()
            m.set_pose(pos, q)
            if bins is None:
                m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0])
            else:
                m.ogm_multiscan_dev(d_pts[i].data_ptr(), bins, rings, 2.0 * math.pi / bins, -math.pi, math.radians(phi_inc), math.radians(phi_min))
            m.step()
            m.sync()
            t1 = time.perf_counter()
            upd.append(1e3 * (t1 - t0))
            if track:
                k, b, n = m.stream_changed()
                strm.append(1e3 * (time.perf_counter() - t1))
                nblk.append(int(n))
        m.close()
        out["track%d" % track] = {"update_ms": [round(v, 3) for v in upd]}
        if track:
            out["track1"]["stream_ms"] = [round(v, 3) for v in strm]
            out["track1"]["blocks"] = nblk
            gb = [n * 512 * 20 / 1e9 for n in nblk]
            out["track1"]["stream_GBps"] = [round(g / (t / 1e3), 2) if t > 0 else None for g, t in zip(gb, strm)]
    out["config"] = {"size": size, "voxel": args.voxel, "sensor": args.sensor}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
