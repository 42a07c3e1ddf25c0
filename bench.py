#!/usr/bin/env python3
"""bench.py — map-update throughput of the MI355X incremental-EDT path.

One step = one full map update of the 512^3 local volume at 0.05 m voxels (set_pose →
OGM of a synthetic lidar point cloud → block alloc + fuse → batch EDT → Mark /
frontiers / waves A,B,C / commit), i.e. VOLMAPNODE::publishMap's GPU work
(src/volumetric_mapper.cpp:138-224).  Sensor frames are generated on the host beforehand and are
resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

With N > 1 every rank owns one 512^3 tile of a larger volume around the same robot (2x2x2 tiles =
1024^3 on 8 GPUs): same sensor stream, one-voxel halo exchange + refinement rounds over RCCL
point-to-point after every update (gie/tiling.py).  Per-GPU work is fixed: scaling is "weak".
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "gie-mapping_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec

# algorithmic bytes per voxel per launch (reference field widths)
ALG_BYTES = {
    "ogm_classify": 1, "ray_finalize": 5, "fuse": 15, "edt_pass_y": 9, "edt_pass_x": 16, "edt_pass_z": 16,
    "mark": 33, "frontiers": 13, "commit": 37,
}
EDT_UPDATE_BYTES = 124  # V3..V8


# sensor models: name -> (rings, azimuth steps of the synthetic cloud, phi_min_deg, phi_inc_deg, range-image bins or None)
SENSORS = {
    # DEFAULT — BASELINE config "UGV VLP-16 3D LiDAR (ugv_dataset), 512^3 local volume, full wavefront A+B":
    # launch/ugv_dataset.launch sets data_case=ugv_corridor, i.e. the point cloud goes through
    # PntcldMapMaker → PNTCLD_RAYCAST (parallel ray casting).  Synthetic 16-ring x 1800 cloud.
    "vlp16": (16, 1800, -15.0, 2.0, None),
    # the same cloud through the projective path of launch/ugv_laser3d.launch (data_case=laser3D):
    # convertPyntCld binning → MulScanParam(440,16,10,2pi/440,-pi,2deg,-15deg) (volumetric_mapper.cpp:327)
    # → VLP_FAST.  Classifies every voxel of the +-15 deg wedge: dense maps, heavy wavefronts.
    "vlp16_projective": (16, 1800, -15.0, 2.0, 440),
    # a denser 64-ring unit through both paths
    "lidar64": (64, 1800, -30.0, 60.0 / 64, None),
    "lidar64_projective": (64, 1800, -30.0, 60.0 / 64, 1800),
}


def make_frames(scenes, voxel, nframes, seed, sensor, delta_vox=8, yaw_deg=2.0, offset=(0.0, 0.0, 0.0)):
    rings, az, phi_min, phi_inc, bins = SENSORS[sensor]
    world = scenes.BoxWorld(seed, extent=(12.0, 12.0, 3.0), n_boxes=200, toggle_frac=0.25, ground_z=-1.5,
                            min_size=0.4, max_size=3.0)
    out = []
    for k in range(nframes):
        pos, q = scenes.pose(k, voxel, delta_vox=delta_vox, yaw_deg=yaw_deg)
        pos = tuple(np.float32(pos[i] + offset[i]) for i in range(3))
        pts, _ = scenes.lidar_frame(world, k, pos, q, rings=rings, az=az, phi_min_deg=phi_min, phi_inc_deg=phi_inc,
                                    max_range=30.0)
        npts = pts.shape[0]
        if bins is not None:   # Vlp16MapMaker::convertPyntCld binning of the cloud (vlp16_map_maker.cpp:73-147)
            pts = scenes.range_image(pts, scan_num=bins, ring_num=rings, phi_min_deg=phi_min, phi_inc_deg=phi_inc)
        out.append((pos, q, pts, npts))
    return out


def cpu_baseline(scenes, voxel, cutoff_dist, sensor):
    """The CPU oracle (a scalar port of the reference's algorithm) on a bounded sample of the
    same workload: same scene generator / sensor, 256^3 local grid, 24 map updates (≈ 12 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_py import OracleMapper
    import gie
    size = (256, 256, 256)
    frames = make_frames(scenes, voxel, 24, 5, sensor)
    rings, az, phi_min, phi_inc, bins = SENSORS[sensor]
    cfg = gie.make_config(voxel, size, cutoff_dist=cutoff_dist, fast_mode=False)
    m = OracleMapper(cfg)
    t0 = time.perf_counter()
    for pos, q, pts, _ in frames:
        if bins is None:
            m.update(pos, q, "pointcloud", pts)
        else:
            m.update(pos, q, "multiscan", pts, theta_inc=2.0 * math.pi / bins, theta_min=-math.pi,
                     phi_inc=math.radians(phi_inc), phi_min=math.radians(phi_min))
    dt = time.perf_counter() - t0
    m.close()
    n = size[0] * size[1] * size[2] * len(frames)
    return {"value": round(n / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": "256^3 local grid, same scene/%s generator, %d map updates (%.1f s)" % (sensor, len(frames), dt)}


def timed_updates(torch, dist, m, step, warmup, steps, instrumented):
    """W untimed map updates, then exactly K timed ones between barrier + synchronize on both sides.
    instrumented = per-kernel HIP events on the mapper's stream: every timed kernel then carries a
    completion signal the next dispatch waits for (≈ +0.1 ms per map update), so the throughput is
    taken from an uninstrumented pass and the kernel durations from an instrumented pass over the
    same map updates on a fresh mapper."""
    for i in range(warmup):
        step(m, i)
    m.sync()
    st0 = m.stats()
    m.profile_enable(bool(instrumented))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        step(m, i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    m.sync()  # surfaces device-side capacity errors
    return dt, st0


def secondary_run(gie, scenes, torch, dev, sensor, size, voxel, cutoff_dist, warmup, steps):
    """The same map update on the dense-observation preset (range-image OGM: most of the volume
    becomes known and waves A/B/C flood), reported beside the headline so that the wavefront
    kernels are seen under load.  Same timing rules as the main run."""
    rings, az, phi_min, phi_inc, bins = SENSORS[sensor]
    frames = make_frames(scenes, voxel, warmup + steps, 5, sensor)
    d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
    cfg = gie.make_config(voxel, size, cutoff_dist=cutoff_dist, fast_mode=False, device_id=dev.index or 0)

    def step(m, i):
        m.set_pose(frames[i][0], frames[i][1])
        m.ogm_multiscan_dev(d_pts[i].data_ptr(), bins, rings, 2.0 * math.pi / bins, -math.pi, math.radians(phi_inc), math.radians(phi_min))
        m.step()

    m = gie.Mapper(cfg)
    dt, st0 = timed_updates(torch, None, m, step, warmup, steps, False)
    st = m.stats()
    known = float((m.read_local(edt=False, dist_sq=False, coc=False)["type"] != 0).mean())
    m.close()
    m = gie.Mapper(cfg)                                   # the same map updates again, with per-kernel events
    timed_updates(torch, None, m, step, warmup, steps, True)
    prof = {k: v for k, v in m.profile_read().items() if v[1] > 0}
    m.close()
    n_vox = size[0] * size[1] * size[2]
    visits = {k: (st["total_visits_" + k] - st0["total_visits_" + k]) / float(steps) for k in "abc"}
    wave_ms = sum(prof[k][0] for k in ("waves",) if k in prof) / steps
    return {"sensor": sensor, "ms_per_step": round(1e3 * dt / steps, 4), "hz": round(steps / dt, 3),
            "value": round(n_vox * steps / dt / 1e6, 2), "unit": "Mvoxels/s", "known_voxel_fraction": round(known, 4),
            "wave_visits_per_step": [round(visits[k], 1) for k in "abc"],
            "wave_ms_per_step": round(wave_ms, 4),
            "wave_visit_rate_Mvisits_per_s": round(sum(visits.values()) / (wave_ms * 1e-3) / 1e6, 1) if wave_ms > 0 else None,
            "edt_update_frac_of_hbm_peak": round(EDT_UPDATE_BYTES * n_vox * (steps / dt) / (HBM_PEAK_GBS * 1e9), 4),
            "kernels_ms_per_step": {k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, nargs=3, default=[512, 512, 512])
    ap.add_argument("--voxel", type=float, default=0.05)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the dense-observation run reported beside the headline")
    ap.add_argument("--sensor", choices=sorted(SENSORS), default="vlp16")
    args = ap.parse_args()

    import torch
    import gie
    from gie import scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    # GIE_BENCH_BACKEND=gloo (+ GIE_BENCH_SHARE_GPU=1: every rank on cuda:0) is a functional check of the
    # N > 1 path on a one-GPU box: face layers are staged through the host.  The measured path is "nccl".
    backend = os.environ.get("GIE_BENCH_BACKEND", "nccl")
    if os.environ.get("GIE_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    size = tuple(args.size)
    n_vox = size[0] * size[1] * size[2]
    cutoff_dist = 2.0
    rings, az, phi_min, phi_inc, bins = SENSORS[args.sensor]
    ALG_BYTES["fuse"] = 15 if bins is None else 7      # ray-cast fuse also reads/zeroes _ray_count
    nframes = args.warmup + args.steps
    # rank r maps tile r of a block-aligned arrangement of 512^3 tiles (2x2x2 = 1024^3 on 8 GPUs):
    # ONE robot / sensor stream shared by all ranks, each rank's local volume offset to its tile,
    # one-voxel halo exchange + refinement rounds over RCCL after every map update
    from gie import tiling
    frames = make_frames(scenes, args.voxel, nframes, 5, args.sensor)
    tgrid = tiling.tile_grid(world)
    whole = tuple(tgrid[i] * size[i] for i in range(3))
    dev = torch.device("cuda", local_rank)
    d_pts = [torch.from_numpy(f[2]).to(dev) for f in frames]
    torch.cuda.synchronize()

    cfg = gie.make_config(args.voxel, size, cutoff_dist=cutoff_dist, fast_mode=False, device_id=local_rank)
    m = gie.Mapper(cfg)
    halo_bufs = {}
    if world > 1:
        m.set_tile(tiling.tile_offset_voxels(rank, world, size), whole)
    rounds_total = [0]
    hr = os.environ.get("GIE_HALO_ROUNDS", "1")
    halo_mode = ["stable" if hr == "stable" else "stream"]
    halo_rounds = 1 if hr == "stable" else max(1, int(hr))
    ray_cells = [None]

    exchange = [world > 1]

    def step(m, i):
        pos, q = frames[i][0], frames[i][1]
        m.set_pose(pos, q)
        if bins is None:
            m.ogm_pointcloud_dev(d_pts[i].data_ptr(), d_pts[i].shape[0])
        else:
            m.ogm_multiscan_dev(d_pts[i].data_ptr(), bins, rings, 2.0 * math.pi / bins, -math.pi,
                                math.radians(phi_inc), math.radians(phi_min))
        if ray_cells[0] is None and bins is None and rank == 0:
            # once, in the warm-up: how many cells does a scan's ray casting count in?  (|_ray_count| summed over
            # the volume: every hit and every cleared cell is one visit = the algorithmic unit of the ray kernels)
            ray_cells[0] = int(np.abs(m.read_ogm()["ray_count"].astype(np.int64)).sum())
        m.step()
        if exchange[0]:
            if backend != "nccl":
                rounds_total[0] += tiling.exchange_until_stable(m, dist, rank, world)
            elif halo_mode[0] == "stream":
                # one exchange round per map update, enqueued on the mapper's own stream (RCCL included): the host never waits.
                # Information crosses one tile boundary per map update.  GIE_HALO_ROUNDS=stable: rounds until no tile changes.
                try:
                    rounds_total[0] += tiling.exchange_rounds_device(m, dist, rank, world, dev, halo_bufs, rounds=halo_rounds)
                except Exception as e:                                  # e.g. no external-stream support: host-synchronised rounds
                    sys.stderr.write("bench: stream-ordered exchange failed (%s); falling back to synchronised rounds\n" % e)
                    halo_mode[0] = "stable"
                    rounds_total[0] += tiling.exchange_until_stable_device(m, dist, rank, world, dev, halo_bufs)
            else:
                rounds_total[0] += tiling.exchange_until_stable_device(m, dist, rank, world, dev, halo_bufs)

    dt, st0 = timed_updates(torch, dist, m, step, args.warmup, args.steps, False)
    st = m.stats()
    known = None
    units = {}
    if rank == 0:
        ty = m.read_local(edt=False, dist_sq=False, coc=False)["type"]
        kn = ty != 0
        known = float(kn.mean())
        # the units a launch processes (per-unit bytes x units of one launch): the sweeps only work on
        # observed voxels, the EDT passes Y / X on the planes that hold obstacles, pass Z on the tiles Mark reads
        n_known = int(kn.sum())
        planes = int((ty == 2).any(axis=(1, 2)).sum())
        Zs, Ys, Xs = ty.shape
        pad = [(0, (-Zs) % 8), (0, (-Ys) % 8), (0, (-Xs) % 8)]
        kt = np.pad(kn, pad).reshape((Zs + pad[0][1]) // 8, 8, (Ys + pad[1][1]) // 8, 8, (Xs + pad[2][1]) // 8, 8).any(axis=(1, 3, 5))
        units = {"fuse": n_known, "mark": n_known, "frontiers": n_known, "commit": n_known,
                 "edt_pass_y": planes * Ys * Xs, "edt_pass_x": planes * Ys * Xs, "edt_pass_z": int(kt.sum()) * 512,
                 "ogm_classify": n_vox}
        del ty, kn, kt
    m.close()
    prof, dt_instr = {}, None
    if rank == 0:
        # the same map updates once more on a fresh mapper, with per-kernel HIP events on its stream (rank 0's
        # tile; no halo exchange: the exchange kernels are not roofline candidates)
        exchange[0] = False
        m2 = gie.Mapper(cfg)
        if world > 1:
            m2.set_tile(tiling.tile_offset_voxels(rank, world, size), whole)
        dt_instr, _ = timed_updates(torch, None, m2, step, args.warmup, args.steps, True)
        prof = m2.profile_read()
        m2.close()

    t_max = dt
    if dist is not None:
        t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_max = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * t_max / args.steps
        hz = args.steps / t_max
        value = world * n_vox * args.steps / t_max / 1e6
        # dominant kernel of the timed map updates (HIP events on the mapper's own stream, instrumented pass)
        sweeps = {k: v for k, v in prof.items() if v[1] > 0}
        total_kernel_ms = sum(v[0] for v in sweeps.values())
        dom = max(sweeps, key=lambda k: sweeps[k][0])
        dom_ms = sweeps[dom][0] / sweeps[dom][1]
        roof = None
        if dom in ALG_BYTES:
            achieved = ALG_BYTES[dom] * units.get(dom, n_vox) / (dom_ms * 1e-3) / 1e9
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tp):
                try:
                    traffic = sum(v for k3, v in json.load(open(tp)).items() if k3 == dom or k3.startswith(dom + ".")) or None
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_ms": round(dom_ms, 4), "alg_bytes_per_voxel": ALG_BYTES[dom], "voxels_per_launch": units.get(dom, n_vox),
                    "note": "achieved = the reference's per-voxel bytes x the voxels one launch works on (observed voxels for the "
                            "sweeps, the planes that hold obstacles for EDT passes Y/X, the tiles Mark reads for pass Z; state of "
                            "the last timed map update) / average launch time; traffic = rocprofv3 PMC bytes per launch"}
        elif dom == "waves":
            # BFS waves A+B+C (one launch): algorithmic bytes = 64 B per visited voxel (own record + six 8-byte RMWs)
            visits = sum(st["total_visits_" + k] - st0["total_visits_" + k] for k in "abc") / float(sweeps[dom][1])
            achieved = 64.0 * visits / (dom_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_ms": round(dom_ms, 4),
                    "alg_bytes_per_visit": 64, "visits_per_launch": round(visits, 1),
                    "note": "level-synchronous BFS over %.0f voxels per launch on average: bound by the dependent "
                            "cross-XCD round trips of each level (grid barrier), not by HBM bandwidth" % visits}
        elif dom in ("ray_free", "ray_register") and ray_cells[0]:
            # ray casting: algorithmic bytes = 13 B per visited cell (1 B scan label read + one 4-byte atomic
            # read-modify-write + its 4-byte return path + 4 B of the ray's own state amortised), SURVEY §8d row R
            achieved = 13.0 * ray_cells[0] / (dom_ms * 1e-3) / 1e9
            traffic = None
            tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tp):
                try:
                    traffic = sum(v for k3, v in json.load(open(tp)).items() if k3 == dom or k3.startswith(dom + ".")) or None
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_launch_ms": round(dom_ms, 4),
                    "alg_bytes_per_cell": 13, "cells_per_launch": ray_cells[0],
                    "note": "per-ray 3-D DDA: a chain of dependent cell visits with one atomic each; bound by that latency chain "
                            "(8 segments of a ray run side by side, 64 adjacent rays share atomics), not by HBM bandwidth"}
        else:
            roof = {"bound": "hbm", "kernel": dom, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                    "traffic": None, "avg_launch_ms": round(dom_ms, 4),
                    "note": "dominant kernel is not volume-proportional"}
        # every volume sweep against the same roofline (algorithmic bytes of SURVEY §8d)
        sweeps_roof = {}
        for k2, v2 in sweeps.items():
            if k2 in ALG_BYTES:
                ms2 = v2[0] / v2[1]
                u2 = units.get(k2, n_vox)
                sweeps_roof[k2] = {"avg_launch_ms": round(ms2, 4), "voxels_per_launch": u2,
                                   "achieved_GBps": round(ALG_BYTES[k2] * u2 / (ms2 * 1e-3) / 1e9, 1),
                                   "frac": round(ALG_BYTES[k2] * u2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # HBM bytes one map update really moves: rocprofv3 PMC (FETCH_SIZE corrected + WRITE_SIZE) per launch of the
        # committed profile of this workload, summed over the kernels of a step (None without the profile)
        measured_bytes = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if args.sensor == "vlp16" and tuple(args.size) == (512, 512, 512) and os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                measured_bytes = tj.get("_per_step_total_bytes")
            except Exception:
                measured_bytes = None
        line = {
            "metric": "edt_map_update_throughput", "value": round(value, 2), "unit": "Mvoxels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "hz": round(hz, 3),
            "config": {"workload": "%dx%dx%d local grid @ %.2f m, synthetic %d-ring x %d lidar point cloud (%d pts/frame) via %s, "
                                   "OGM + fuse + batch EDT + waves A/B/C + commit, cutoff %.1f m"
                                   % (size[0], size[1], size[2], args.voxel, rings, az, int(np.mean([f[3] for f in frames])),
                                      "parallel ray casting" if bins is None else "%dx%d range image (projective OGM)" % (rings, bins),
                                      cutoff_dist),
                       "sensor": args.sensor,
                       "tiles": ("%dx%dx%d tiles of %dx%dx%d, one per GPU, one-voxel halo exchange + refinement (%.1f rounds/step, %s)"
                                 % (tgrid + size + (rounds_total[0] / float(nframes),
                                                    "stream-ordered, fixed" if (halo_mode[0] == "stream" and backend == "nccl") else "until no tile changes"))) if world > 1 else "single volume",
                       "known_voxel_fraction": known,
                       "wave_visits_per_step": [round((st["total_visits_" + k] - st0["total_visits_" + k]) / args.steps, 1) for k in "abc"],
                       "wave_levels_last_step": [st["levels_a"], st["levels_b"], st["levels_c"]],
                       "blocks": st["blocks_total"]},
            "edt_update_frac_of_hbm_peak": round(EDT_UPDATE_BYTES * n_vox * hz / (HBM_PEAK_GBS * 1e9), 4),
            "edt_update_frac_note": "BASELINE.md's convention: the reference's 124 B per voxel of the WHOLE volume x Hz / 8 TB/s; above 1 "
                                    "because the kernels only touch observed space (see measured_hbm_bytes_per_step)",
            "measured_hbm_bytes_per_step": measured_bytes,
            "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in sorted(sweeps.items(), key=lambda kv: -kv[1][0])},
            "kernel_time_fraction_of_step": round(total_kernel_ms / (1e3 * dt), 3),
            "ms_per_step_instrumented": round(1e3 * dt_instr / args.steps, 4),
            "instrumentation_note": "value / ms_per_step: the K timed map updates without per-kernel events; kernels_ms_per_step, roofline: "
                                    "the same K map updates replayed on a fresh mapper with start/stop events on every kernel's dispatch "
                                    "(hipExtLaunchKernelGGL, the mapper's stream), which costs ms_per_step_instrumented - ms_per_step",
            "roofline": roof,
            "roofline_sweeps": sweeps_roof,
        }
        if world == 1 and args.sensor == "vlp16" and not args.no_secondary:
            line["dense_observation_run"] = secondary_run(gie, scenes, torch, dev, "vlp16_projective", size, args.voxel, cutoff_dist,
                                                          args.warmup, args.steps)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scenes, args.voxel, cutoff_dist, args.sensor)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()                                    # rank 0's instrumented pass is over
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
