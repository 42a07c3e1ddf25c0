#!/usr/bin/env python3
"""bench.py — map-update throughput of the MI355X incremental-EDT path.

One step = one full map update of a local volume (set_pose → OGM → block allocation + fuse → batch EDT → Mark /
obtainFrontiers / waves A,B,C / commit), i.e. the GPU work of VOLMAPNODE::publishMap (src/volumetric_mapper.cpp:138-224).
Inputs are resident in HBM when a timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c5|c2|c3|c4|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (all synthetic; PRESETS below holds grid / voxel / cutoff / fast_mode of each):
  c5 (default, headline)  BASELINE config 5's sensor-less hash world on a 512^3 grid at 0.05 m, cutoff 2 m: FULL observation,
                     a quarter of the obstacles toggles every frame, the robot moves 8 voxels per frame.
  c2 / c2_projective BASELINE config 2: 256^3 at 0.05 m, cutoff 2 m, 640x480 depth camera; the 307 200-point cloud through
                     parallel ray casting (the reference's cow_lady path) / the same image through the projective depth kernel.
  c3 / c3_projective BASELINE config 3: 512^3 at 0.1 m, cutoff 100 m (= no cutoff), fast_mode off, 16-ring lidar with 100 m
                     range; ray casting / the 16x440 range image through the projective lidar kernel.
  c4 / c4_nofast     BASELINE config 4 (uav_raycast): 320x320x40 at 0.05 m, cutoff 5 m, lidar through ray casting,
                     fast_mode on (cfg/uav_laser3D_params.yaml:27) / off.
  vlp16* / lidar64*  a lidar in the headline's 512^3 / 0.05 m / 2 m volume (ray casting, or *_projective).
With N > 1 every rank owns one 512^3 tile of a larger volume around the same robot (2x2x2 tiles = 1024^3 on 8 GPUs =
BASELINE config 5 itself): one-voxel halo exchange + refinement over RCCL after every update until no tile changed
(gie/tiling.py exchange_converged_device).  Per-GPU work is fixed: scaling is "weak".

Rank 0 prints ONE JSON line, kept below 8 KB (build_line): headline keys, ONE roofline object, cpu_baseline, per-workload
step times.  Everything else (per-kernel roofline trees, notes, regions) goes to profiles/bench_last_full.json.

Timing: W warm-up steps, then regions of EXACTLY K steps, each bracketed by barrier + synchronize on both sides (MAX over
ranks); regions repeat until >= 0.5 s has been timed and `value` / `ms_per_step` come from the median region.  Per-step
latencies (median, p95) come from a further pass with one event pair per step on the mapper's stream, kernel durations from
a replay of the first region on a fresh mapper with HIP events on every kernel's dispatch.
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured with a float4 copy)

# algorithmic bytes per voxel per launch (reference field widths)
ALG_BYTES = {
    "ogm_classify": 1, "fuse": 7, "edt_pass_y": 9, "edt_pass_x": 16, "edt_pass_z": 16,
    "mark": 33, "frontiers": 13, "commit": 37,
    "mark_commit": 33 + 37,    # Mark and commit as one sweep (rows V6 + V8)
}
# bytes per unit THIS build's layout has to move at least ("physical B/voxel touched"): what `frac` is priced
# on when no PMC profile of this exact build is committed (a lower bound of the kernel's real traffic, so frac stays <= 1)
LAYOUT_BYTES = {
    "ogm_classify": 1, "fuse": 6, "edt_pass_y": 3, "edt_pass_x": 6, "edt_pass_z": 8,
    "mark": 25, "frontiers": 9, "commit": 25, "mark_commit": 13,    # (round 4: `_edt_D` is derived from the pairs, no 4-byte store; round 5: the stored
                                                                    #  obstacle of a tskip tile's voxels is left to the pair plane: type 1 + batch obstacle 4 + pair 8;
                                                                    #  round 6: only the tiles on the volume's faces have to be swept at all, kernel_roof())
}
WAVE_VISIT_BYTES = 64      # own record + six 8-byte read-modify-writes
RAY_CELL_BYTES = 13        # row R: 1 B label read + 4 B atomic + 4 B return + ray state amortised
STAGE_LAUNCHES = {"waves": 2, "frontiers": 2, "block_alloc": 7}     # stages of gie_profile_read that are more than one kernel launch (gie_hip.hip)
WAVEFRONT_SWEEP = ("mark", "mark_commit", "frontiers", "waves", "commit")   # GlbHashMap::mergeNewObsv, glb_hash_map.cu:146-207

# lidar models: name -> (rings, azimuth steps, phi_min_deg, phi_inc_deg, range-image bins or None = ray casting)
LIDARS = {
    "vlp16": (16, 1800, -15.0, 2.0, None),
    "vlp16_projective": (16, 1800, -15.0, 2.0, 440),
    "lidar64": (64, 1800, -30.0, 60.0 / 64, None),
    "lidar64_projective": (64, 1800, -30.0, 60.0 / 64, 1800),
}
# name -> grid, voxel [m], cutoff [m], fast_mode, feed ("hash" | "lidar" | "depth"), lidar/depth model, robot step [voxels/frame],
# box world (seed, half extent [m], boxes, min / max box size [m]), sensor range [m]
PRESETS = {
    "c5": {"size": (512, 512, 512), "voxel": 0.05, "cutoff": 2.0, "fast": False, "feed": "hash", "delta": 8},
    "c2": {"size": (256, 256, 256), "voxel": 0.05, "cutoff": 2.0, "fast": False, "feed": "depth", "projective": False, "delta": 4,
           "world": (2, (8.0, 8.0, 3.0), 80, 0.4, 2.5), "range": 8.0},
    "c2_projective": {"size": (256, 256, 256), "voxel": 0.05, "cutoff": 2.0, "fast": False, "feed": "depth", "projective": True, "delta": 4,
                      "world": (2, (8.0, 8.0, 3.0), 80, 0.4, 2.5), "range": 8.0},
    "c3": {"size": (512, 512, 512), "voxel": 0.1, "cutoff": 100.0, "fast": False, "feed": "lidar", "sensor": "vlp16", "delta": 8,
           "world": (3, (40.0, 40.0, 4.0), 300, 1.0, 6.0), "range": 100.0},
    "c3_projective": {"size": (512, 512, 512), "voxel": 0.1, "cutoff": 100.0, "fast": False, "feed": "lidar", "sensor": "vlp16_projective",
                      "delta": 8, "world": (3, (40.0, 40.0, 4.0), 300, 1.0, 6.0), "range": 100.0},
    "c4": {"size": (320, 320, 40), "voxel": 0.05, "cutoff": 5.0, "fast": True, "feed": "lidar", "sensor": "vlp16", "delta": 4,
           "world": (4, (12.0, 12.0, 3.0), 200, 0.4, 3.0), "range": 30.0},
    "c4_nofast": {"size": (320, 320, 40), "voxel": 0.05, "cutoff": 5.0, "fast": False, "feed": "lidar", "sensor": "vlp16", "delta": 4,
                  "world": (4, (12.0, 12.0, 3.0), 200, 0.4, 3.0), "range": 30.0},
}
for _name in LIDARS:     # a lidar in the headline's volume (rounds 1-4's extra runs)
    PRESETS[_name] = {"size": (512, 512, 512), "voxel": 0.05, "cutoff": 2.0, "fast": False, "feed": "lidar", "sensor": _name, "delta": 8,
                      "world": (5, (12.0, 12.0, 3.0), 200, 0.4, 3.0), "range": 30.0}
WORKLOADS = list(PRESETS)
BASELINE_CONFIG = {"c5": 5, "c2": 2, "c2_projective": 2, "c3": 3, "c3_projective": 3, "c4": 4, "c4_nofast": 4}
DEFAULT_EXTRAS = ("vlp16_projective", "vlp16", "c2", "c2_projective", "c3", "c3_projective", "c4", "c4_nofast")
DEPTH_CAM = {"rows": 480, "cols": 640, "fx": 525.0, "fy": 525.0, "cx": 319.5, "cy": 239.5}
C5 = {"seed": 5, "p_occ": 0.01, "toggle_frac": 0.25, "delta_vox": 8, "yaw_deg": 2.0}
C5_TURN = 24             # like the lidar robot below, the c5 robot drives 24 frames out (+8 voxels each) and 24 back, so the map it
                         # leaves behind is bounded however many regions a command times (round 2: a straight line forever ran
                         # the default block pool dry after ~157 updates and the driver's --steps 20 --warmup 5 died in it)
MAX_REGIONS = 12         # timed regions per workload at most
HALO_MAX_ROUNDS = 3      # N > 1: refinement rounds enqueued per map update at most (the tail rounds exit on the device-side flag).  The hash world
                         # needs 2 - 3 on 2x2x2 tiles whatever the tile size (the tiled oracle on 24^3 and 64^3 tiles: 2, 2, 3, 2) — the last of them
                         # is the round that seeds nothing; an update that would need more is counted (config.updates_unconverged).  A round that
                         # meets a closed gate still costs its launches and its transfers (a collective cannot be skipped by one side): ~0.15 ms


SENSORS = LIDARS


def box_world(scenes, spec):
    seed, extent, n_boxes, lo, hi = spec
    return scenes.BoxWorld(seed, extent=extent, n_boxes=n_boxes, toggle_frac=0.25, ground_z=-1.5, min_size=lo, max_size=hi)


def lidar_world(scenes):
    return box_world(scenes, PRESETS["vlp16"]["world"])


LIDAR_TURN = 24          # the robot of the lidar workloads drives 24 frames out and 24 back: it stays inside the 12 m box world
                         # however many regions are timed


def turn_index(i, turn):
    """Frame i of an out-and-back drive: 0, 1, …, turn, turn-1, …, 1, 0, 1, …"""
    j = i % (2 * turn)
    return j if j <= turn else 2 * turn - j


DRIVE = {"mode": "turn", "retain": 0}      # --drive line: the c5 robot never turns back (needs --retain R or a pool for the whole drive)


def c5_pose(scenes, i, voxel):
    return scenes.pose(turn_index(i, C5_TURN) if DRIVE["mode"] == "turn" else i, voxel, delta_vox=C5["delta_vox"], yaw_deg=C5["yaw_deg"])


def planned_updates(W, K, max_regions=MAX_REGIONS, with_latency=True):
    """Upper bound of the map updates one mapper of run_workload() goes through: warm-up + every timed region + the latency pass."""
    return W + K * max_regions + (K if with_latency else 0)


def c5_pool_blocks(size, updates):
    """Blocks the c5 drive can allocate in `updates` map updates (an upper bound: every 8x8x8 block the volume ever overlaps,
    +1 per axis for a pivot that is not block-aligned, +1 for the ghost layer of a tiled run)."""
    if DRIVE["retain"] > 0:             # block-pool lifecycle on: the retention zone is all the map ever holds
        n = 1
        for e in size:
            n *= (e + 7) // 8 + 2 + 2 * DRIVE["retain"]
        return n + ((size[1] + 7) // 8 + 2) * ((size[2] + 7) // 8 + 2) * ((C5["delta_vox"] + 7) // 8)   # + what one step adds before the next erasure
    travel = C5["delta_vox"] * (min(int(updates), C5_TURN) if DRIVE["mode"] == "turn" else int(updates))
    ext = (size[0] + travel, size[1], size[2])
    n = 1
    for e in ext:
        n *= (e + 7) // 8 + 2
    return n


def pool_blocks(workload, size, updates):
    """gie_config.max_blocks for a bench run of `updates` map updates (0 = the library's default of 3 x the block table)."""
    if workload != "c5":
        return 0
    return int(c5_pool_blocks(size, updates) * 1.05) + 4096


def lidar_host_frame(scenes, world, voxel, sensor, i, delta_vox=8, max_range=30.0, device=None):
    """(pos, quat, cloud or range image, points in the cloud) of frame i of a lidar workload."""
    rings, az, phi_min, phi_inc, bins = LIDARS[sensor]
    pos, q = scenes.pose(turn_index(i, LIDAR_TURN), voxel, delta_vox=delta_vox, yaw_deg=2.0)
    pts, _ = scenes.lidar_frame(world, i, pos, q, rings=rings, az=az, phi_min_deg=phi_min, phi_inc_deg=phi_inc, max_range=max_range, device=device)
    npts = pts.shape[0]
    if bins is not None:   # Vlp16MapMaker::convertPyntCld binning (vlp16_map_maker.cpp:73-147)
        pts = scenes.range_image(pts, scan_num=bins, ring_num=rings, phi_min_deg=phi_min, phi_inc_deg=phi_inc)
    return pos, q, pts, npts


def make_frames(scenes, voxel, nframes, seed, sensor):
    """Host-side frames of a lidar workload (tests replay the bench's scenes through the oracle)."""
    world = lidar_world(scenes)
    return [lidar_host_frame(scenes, world, voxel, sensor, i) for i in range(nframes)]


class HashWorldFeed:
    """BASELINE config 5: label planes of the hash world, built on the device between timed regions."""

    kind = "labels"

    def __init__(self, torch, scenes, dev, voxel, size, tile_off):
        self.torch, self.scenes, self.dev, self.voxel, self.size, self.tile_off = torch, scenes, dev, voxel, size, tile_off
        self.first, self.planes = 0, []

    def pose(self, i):
        return c5_pose(self.scenes, i, self.voxel)

    def prepare(self, first, count):
        torch = self.torch
        while len(self.planes) < count:
            self.planes.append(torch.empty(self.size[2], self.size[1], self.size[0], dtype=torch.int8, device=self.dev))
        for j in range(count):
            pos, _ = self.pose(first + j)
            pvt = self.scenes.local_pivot(pos, self.voxel, self.size, self.tile_off)
            lab = self.scenes.hash_world_labels(
                pvt, self.size, first + j, seed=C5["seed"], p_occ=C5["p_occ"], toggle_frac=C5["toggle_frac"],
                arange=lambda n: torch.arange(n, dtype=torch.int64, device=self.dev),
                where=lambda c, a, b: torch.where(c, torch.tensor(a, dtype=torch.int8, device=self.dev), torch.tensor(b, dtype=torch.int8, device=self.dev)))
            self.planes[j].copy_(lab)
            del lab
        self.first = first
        torch.cuda.synchronize()

    def step_input(self, m, i):
        pos, q = self.pose(i)
        m.set_pose(pos, q)
        m.ogm_labels_dev(self.planes[i - self.first].data_ptr(), borrow=True)      # (the planes stay resident and untouched: gie_fuse reads them in place)

    def describe(self):
        return ("sensor-less hash world (BASELINE config 5): occupied iff hash(x,y,z) < %.0f %%, full observation, %.0f %% of the "
                "obstacles toggle per frame, robot %d voxels/frame, %s" % (100 * C5["p_occ"], 100 * C5["toggle_frac"], C5["delta_vox"],
                   ("%d frames out and %d back" % (C5_TURN, C5_TURN)) if DRIVE["mode"] == "turn" else
                   ("a straight line, blocks more than %d blocks behind the volume erased and recycled" % DRIVE["retain"] if DRIVE["retain"] else "a straight line")))


class LidarFeed:
    """A 16- or 64-ring lidar in a box world: point clouds (ray casting) or range images (projective OGM)."""

    def __init__(self, torch, scenes, dev, voxel, sensor, nframes, preset=None):
        self.torch, self.scenes, self.dev, self.voxel, self.sensor = torch, scenes, dev, voxel, sensor
        self.rings, self.az, self.phi_min, self.phi_inc, self.bins = LIDARS[sensor]
        self.kind = "pointcloud" if self.bins is None else "multiscan"
        p = preset or PRESETS[sensor]
        self.delta, self.max_range = p["delta"], p["range"]
        self.world = box_world(scenes, p["world"])
        self.cast_dev = dev if getattr(dev, "type", "cpu") == "cuda" else None      # ray / box tests of the synthetic scene on the GPU
        self.frames = {}
        self.npts = []

    def _frame(self, i):
        if i not in self.frames:
            pos, q, pts, npts = lidar_host_frame(self.scenes, self.world, self.voxel, self.sensor, i, self.delta, self.max_range, self.cast_dev)
            self.npts.append(npts)
            self.frames[i] = (pos, q, pts, self.torch.from_numpy(pts).to(self.dev))
        return self.frames[i]

    def prepare(self, first, count):
        for i in range(first, first + count):
            self._frame(i)
        if self.cast_dev is not None:
            self.torch.cuda.synchronize()

    def step_input(self, m, i):
        pos, q, _, d = self._frame(i)
        m.set_pose(pos, q)
        if self.bins is None:
            m.ogm_pointcloud_dev(d.data_ptr(), d.shape[0])
        else:
            m.ogm_multiscan_dev(d.data_ptr(), self.bins, self.rings, 2.0 * math.pi / self.bins, -math.pi,
                                math.radians(self.phi_inc), math.radians(self.phi_min))

    def oracle_update(self, om, i):
        pos, q, pts, _ = self._frame(i)
        if self.bins is None:
            om.update(pos, q, "pointcloud", pts)
        else:
            om.update(pos, q, "multiscan", pts, theta_inc=2.0 * math.pi / self.bins, theta_min=-math.pi,
                      phi_inc=math.radians(self.phi_inc), phi_min=math.radians(self.phi_min))

    def describe(self):
        return ("%d-ring x %d lidar, %d pts/frame, %s" % (self.rings, self.az, int(np.mean(self.npts)) if self.npts else 0,
                "ray casting" if self.bins is None else "%dx%d range image (projective)" % (self.rings, self.bins)))


class DepthFeed:
    """BASELINE config 2: a 640x480 pinhole depth camera in a box world; the valid pixels as a point cloud through ray casting
    (PntcldMapMaker, the reference's cow_lady launch) or the image itself through the projective depth kernel (RealsenseMapMaker)."""

    def __init__(self, torch, scenes, dev, voxel, preset):
        self.torch, self.scenes, self.dev, self.voxel = torch, scenes, dev, voxel
        self.projective = bool(preset["projective"])
        self.kind = "depth" if self.projective else "pointcloud"
        self.delta, self.max_range = preset["delta"], preset["range"]
        self.world = box_world(scenes, preset["world"])
        self.cast_dev = dev if getattr(dev, "type", "cpu") == "cuda" else None
        self.frames = {}
        self.npts = []

    def _frame(self, i):
        if i not in self.frames:
            pos, q = self.scenes.pose(turn_index(i, LIDAR_TURN), self.voxel, delta_vox=self.delta, yaw_deg=2.0)
            depth = self.scenes.depth_frame(self.world, i, pos, q, max_depth=self.max_range, device=self.cast_dev, **DEPTH_CAM)
            host = depth if self.projective else self.scenes.depth_to_points(depth, DEPTH_CAM["fx"], DEPTH_CAM["fy"], DEPTH_CAM["cx"], DEPTH_CAM["cy"])
            self.npts.append(int(np.isfinite(depth).sum()))
            self.frames[i] = (pos, q, host, self.torch.from_numpy(host).to(self.dev))
        return self.frames[i]

    def prepare(self, first, count):
        for i in range(first, first + count):
            self._frame(i)
        if self.cast_dev is not None:
            self.torch.cuda.synchronize()

    def step_input(self, m, i):
        pos, q, _, d = self._frame(i)
        m.set_pose(pos, q)
        if self.projective:
            m.ogm_depth_dev(d.data_ptr(), DEPTH_CAM["rows"], DEPTH_CAM["cols"], DEPTH_CAM["cx"], DEPTH_CAM["cy"], DEPTH_CAM["fx"], DEPTH_CAM["fy"], valid_nan=True)
        else:
            m.ogm_pointcloud_dev(d.data_ptr(), d.shape[0])

    def oracle_update(self, om, i):
        pos, q, host, _ = self._frame(i)
        if self.projective:
            om.update(pos, q, "depth", host, cx=DEPTH_CAM["cx"], cy=DEPTH_CAM["cy"], fx=DEPTH_CAM["fx"], fy=DEPTH_CAM["fy"], valid_nan=True)
        else:
            om.update(pos, q, "pointcloud", host)

    def describe(self):
        return ("%dx%d depth camera, %d valid px/frame, %s" % (DEPTH_CAM["cols"], DEPTH_CAM["rows"], int(np.mean(self.npts)) if self.npts else 0,
                "projective depth kernel" if self.projective else "cloud through ray casting"))


def make_feed(workload, torch, scenes, dev, voxel, size, tile_off, nframes):
    p = PRESETS[workload]
    if p["feed"] == "hash":
        return HashWorldFeed(torch, scenes, dev, voxel, size, tile_off)
    if p["feed"] == "depth":
        return DepthFeed(torch, scenes, dev, voxel, p)
    return LidarFeed(torch, scenes, dev, voxel, p["sensor"], nframes, preset=p)


def dominant_stage(prof):
    """The stage that certainly holds the update's longest kernel: prof = {stage: (total ms, launches of the stage's timer)}.  A stage
    of several kernel launches (STAGE_LAUNCHES) competes with the lower bound of its longest kernel, stage time / launches."""
    return max(prof, key=lambda k: prof[k][0] / max(prof[k][1], 1) / STAGE_LAUNCHES.get(k, 1))


def percentile(xs, p):
    xs = sorted(xs)
    if not xs:
        return None
    k = (len(xs) - 1) * p
    lo, hi = int(math.floor(k)), int(math.ceil(k))
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


class Runner:
    """One mapper + its feed + (N > 1) the halo exchange after every update."""

    def __init__(self, torch, gie, tiling, dist, feed, cfg, rank, world, size, dev, backend, exchange=True, group=None):
        self.torch, self.dist, self.feed, self.rank, self.world, self.dev, self.backend = torch, dist, feed, rank, world, dev, backend
        self.group = group                                 # the RCCL group of the face-layer exchange (None: the default gloo group)
        self.fallback_note = None
        self.tiling = tiling
        self.m = gie.Mapper(cfg)
        self.exchange = exchange and world > 1
        if world > 1:
            tgrid = tiling.tile_grid(world)
            self.m.set_tile(tiling.tile_offset_voxels(rank, world, size), tuple(tgrid[i] * size[i] for i in range(3)))
        self.halo_bufs = {}
        # GIE_HALO_ROUNDS: "converged" (default) = stream-ordered rounds gated by a device-side "some tile changed" word that is
        # all-reduced over RCCL on the mapper's stream, at most HALO_MAX_ROUNDS of them (tiling.exchange_converged_device: what the
        # parity tests hold against the tiled oracle); "stable" = host-synchronised rounds until no tile changes; an integer = that
        # many ungated stream-ordered rounds (round 4's default was 1)
        hr = os.environ.get("GIE_HALO_ROUNDS", "converged")
        self.halo_mode = hr if hr in ("stable", "converged") else "stream"
        self.halo_rounds = HALO_MAX_ROUNDS if hr in ("stable", "converged") else max(1, int(hr))
        self.rounds_total = 0
        self.updates = 0
        self.checked_pivot = False

    def step(self, i, mid=None):
        """mid: (event, stream) recorded on the mapper's stream between the tile's own work and the face-layer exchange (latency pass)"""
        m = self.m
        self.feed.step_input(m, i)
        if not self.checked_pivot and self.feed.kind == "labels":       # the label planes were built for this pivot
            pos, _ = self.feed.pose(i)
            assert m.pivot() == self.feed.scenes.local_pivot(pos, self.feed.voxel, self.feed.size, self.feed.tile_off), "pivot mismatch"
            self.checked_pivot = True
        if self.exchange:
            m.step_begin_tiled()                        # fuse + batch EDT + first half of the merge; round 0 of the exchange finishes it
        else:
            m.step()
        self.updates += 1
        if mid is not None:
            mid[0].record(mid[1])
        if self.exchange:
            t, d = self.tiling, self.dist
            if self.backend != "nccl":
                self.rounds_total += t.exchange_until_stable(m, d, self.rank, self.world, sparse=os.environ.get("GIE_HALO_SPARSE", "0") == "1")
            elif self.halo_mode in ("stream", "converged"):
                # exchange rounds enqueued on the mapper's own stream (RCCL included): the host never waits
                try:
                    if self.halo_mode == "converged":
                        t.exchange_converged_device(m, d, self.rank, self.world, self.dev, self.halo_bufs, max_rounds=self.halo_rounds, group=self.group)
                    else:
                        self.rounds_total += t.exchange_rounds_device(m, d, self.rank, self.world, self.dev, self.halo_bufs, rounds=self.halo_rounds, group=self.group)
                except Exception as e:                                  # e.g. no external-stream support: host-synchronised rounds
                    self.fallback_note = "stream-ordered exchange failed (%s: %s): host-synchronised rounds" % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else "")
                    sys.stderr.write("bench: %s\n" % self.fallback_note)
                    self.halo_mode = "stable"
                    self.halo_bufs = {}
                    self.rounds_total += t.exchange_until_stable_device(m, d, self.rank, self.world, self.dev, self.halo_bufs, group=self.group, sparse=os.environ.get("GIE_HALO_SPARSE", "0") == "1")
            else:
                self.rounds_total += t.exchange_until_stable_device(m, d, self.rank, self.world, self.dev, self.halo_bufs, group=self.group, sparse=os.environ.get("GIE_HALO_SPARSE", "0") == "1")

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, v):
        if self.dist is None:
            return v
        t = self.torch.tensor([v], device="cpu", dtype=self.torch.float64)       # control plane: the default (gloo) group
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def warmup(self, W):
        self.feed.prepare(0, W)
        for i in range(W):
            self.step(i)
        self.m.sync()
        return W

    def region(self, first, K):
        """EXACTLY K timed steps between barrier + synchronize on both sides; MAX over ranks."""
        self.feed.prepare(first, K)
        self.barrier()
        t0 = time.perf_counter()
        for i in range(first, first + K):
            self.step(i)
        self.barrier()
        dt = time.perf_counter() - t0
        self.m.sync()                                   # surfaces device-side capacity errors
        return self.max_over_ranks(dt)

    def latency_pass(self, first, K):
        """Per-step device time: one event pair per step on the mapper's own stream."""
        torch = self.torch
        self.feed.prepare(first, K)
        s = torch.cuda.ExternalStream(self.m.stream_handle(), device=self.dev)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        self.barrier()
        for j in range(K):
            evs[j][0].record(s)
            self.step(first + j, mid=(evs[j][2], s) if self.exchange else None)
            evs[j][1].record(s)
        self.barrier()
        self.m.sync()
        # N > 1: the part of a step behind the tile's own fuse / EDT / Mark — export, transfers, import, obtainFrontiers + waves, refinement
        # rounds — as far as it is enqueued on the mapper's stream (the stream-ordered RCCL modes; host-staged rounds wait on the host)
        self.exchange_ms = [c.elapsed_time(b) for a, b, c in evs] if self.exchange else []
        return [a.elapsed_time(b) for a, b, c in evs]

    def close(self):
        self.m.close()


def run_workload(torch, gie, scenes, tiling, dist, workload, size, voxel, cutoff_dist, W, K, rank, world, dev, local_rank, backend,
                 min_timed_s=0.5, max_regions=MAX_REGIONS, with_latency=True, rms=False, group=None, transport_note=None, fast_mode=False,
                 default_knobs=False):
    """Timed regions + latency pass + instrumented replay for one workload.  Returns a dict (rank 0) or None."""
    n_vox = size[0] * size[1] * size[2]
    tile_off = tiling.tile_offset_voxels(rank, world, size) if world > 1 else (0, 0, 0)
    cfg = gie.make_config(voxel, size, cutoff_dist=cutoff_dist, fast_mode=fast_mode, device_id=local_rank, retain_radius_blocks=DRIVE["retain"],
                          max_blocks=pool_blocks(workload, size, planned_updates(W, K, max_regions, with_latency)),
                          wave_workgroups=0 if default_knobs else WAVE_GRID["wgs"],
                          place_tries=(PLACE_TRIES if os.environ.get("GIE_BENCH_SHARE_GPU") != "1" else 0) if not default_knobs else 0)
    feed = make_feed(workload, torch, scenes, dev, voxel, size, tile_off, W + K)
    r = Runner(torch, gie, tiling, dist, feed, cfg, rank, world, size, dev, backend, group=group)
    first = r.warmup(W)
    st0 = r.m.stats()
    regions = []
    PARTIAL[workload] = {"n_voxels": n_vox * world, "steps": K, "region_s": regions}    # what a failure later on still reports
    while True:
        regions.append(r.region(first, K))
        first += K
        if sum(regions) >= min_timed_s or len(regions) >= max_regions:
            break
    st1 = r.m.stats()
    timed_updates = K * len(regions)
    lat = r.latency_pass(first, K) if with_latency else []
    first += K if with_latency else 0
    res = None
    accuracy = None
    if rank == 0 and rms:
        accuracy = accuracy_check(r.m, voxel)
    if rank == 0:
        ty = r.m.read_local(edt=False, dist_sq=False, coc=False)["type"]
        kn = ty != 0
        n_known = int(kn.sum())
        planes = int((ty == 2).any(axis=(1, 2)).sum())
        Zs, Ys, Xs = ty.shape
        pad = [(0, (-Zs) % 8), (0, (-Ys) % 8), (0, (-Xs) % 8)]
        kt = np.pad(kn, pad).reshape((Zs + pad[0][1]) // 8, 8, (Ys + pad[1][1]) // 8, 8, (Xs + pad[2][1]) // 8, 8).any(axis=(1, 3, 5))
        # the units one launch works on (per-unit bytes x units per launch): observed voxels for the sweeps,
        # the planes that hold obstacles for EDT passes Y / X, the tiles Mark reads for pass Z; all = N under full observation
        # obtainFrontiers examines the voxels on the six faces of the volume and, voxel by voxel, only the tiles its summary lists
        face = np.zeros_like(kn); face[0] = face[-1] = True; face[:, 0] = face[:, -1] = True; face[:, :, 0] = face[:, :, -1] = True
        n_front = int((kn & face).sum()) + 512 * int(st1.get("frontier_tiles", 0))
        units = {"fuse": n_known, "mark": n_known, "mark_commit": n_known, "frontiers": min(n_known, n_front), "commit": n_known,
                 "edt_pass_y": planes * Ys * Xs, "edt_pass_x": planes * Ys * Xs, "edt_pass_z": int(kt.sum()) * 512,
                 "ogm_classify": n_vox}
        known = n_known / float(n_vox)
        ray_cells = None
        del ty, kn, kt, face
        res = {"known": known, "units": units}
    blocks = st1["blocks_total"]
    rounds_per_step = r.rounds_total / float(max(1, r.updates))
    halo_mode = r.halo_mode
    round_stats = None
    if r.exchange and backend != "nccl":
        halo_mode = "stable"                          # (host-staged layers: rounds until no tile changes, the host reading every count)
    if r.exchange and halo_mode == "converged":      # what the gated rounds did is counted on the device (gie_round_stats)
        round_stats = r.m.round_stats()
        rounds_per_step = round_stats["rounds_run"] / float(max(1, round_stats["updates"]))
        if dist is not None:                          # the worst tile: every rank's count over the control plane
            t = torch.tensor([float(round_stats["rounds_run"]), float(round_stats["updates_unconverged"])], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            round_stats = dict(round_stats, rounds_run=int(t[0].item()), updates_unconverged=int(t[1].item()))
            rounds_per_step = round_stats["rounds_run"] / float(max(1, round_stats["updates"]))
    notes = [n for n in (transport_note, r.fallback_note) if n]
    r.close()
    if rank != 0:
        return None
    # kernel durations: the first region replayed on a fresh mapper with start / stop events on every kernel's dispatch
    # (rank 0's tile, no halo exchange: the exchange kernels are not roofline candidates)
    feed2 = feed            # (same frames: a lidar / depth feed keeps the scans it has generated)
    r2 = Runner(torch, gie, tiling, None, feed2, cfg, rank, world, size, dev, backend, exchange=False)
    r2.warmup(W)
    if feed2.kind == "pointcloud":          # cells one scan's ray casting counts in (hits + cleared cells = the ray kernels' unit)
        feed2.prepare(W, 1)
        feed2.step_input(r2.m, W)
        ray_cells = int(np.abs(r2.m.read_ogm()["ray_count"].astype(np.int64)).sum())
        r2.m.step()
        w0 = W + 1
    else:
        w0 = W
    r2.m.profile_enable(True)
    sv0 = r2.m.stats()
    dt_instr = r2.region(w0, K)
    prof = {k: v for k, v in r2.m.profile_read().items() if v[1] > 0}
    sv1 = r2.m.stats()
    r2.close()

    med = percentile(regions, 0.5)
    ms_per_step = 1e3 * med / K
    hz = K / med
    value = world * n_vox * hz / 1e6
    visits = {k: (st1["total_visits_" + k] - st0["total_visits_" + k]) / float(timed_updates) for k in "abc"}
    visits_instr = sum(sv1["total_visits_" + k] - sv0["total_visits_" + k] for k in "abc") / float(K)
    units = res["units"]
    alg = dict(ALG_BYTES)
    if feed2.kind == "pointcloud":
        alg["fuse"] = 15                     # ray-cast fuse also reads / zeroes _ray_count
    model_errors = []

    def kernel_roof(name):
        tot_ms, n = prof[name]
        avg_ms = tot_ms / n
        sec = avg_ms * 1e-3
        if name in alg:
            u = units.get(name, n_vox)
            b = alg[name] * u
            lay = LAYOUT_BYTES.get(name, alg[name]) * u
            if name == "mark_commit":
                # round 6 ("lazy pairs"): a cleared tile whose records are deferred is not swept at all -- its pairs are derived from the
                # batch-obstacle plane by whoever reads them -- and only a tile on a face of the volume can never be one: what the
                # layout HAS to move is the face tiles' 13 B per voxel (how many tiles are lazy in a given update is the scene's business)
                tiles = [(e + 7) // 8 for e in size]
                inner = 1
                for t_ in tiles:
                    inner *= max(t_ - 2, 0)
                lay = lay * (1.0 - inner / float(tiles[0] * tiles[1] * tiles[2]))
            what = {"alg_bytes_per_voxel": alg[name], "layout_bytes_per_voxel": LAYOUT_BYTES.get(name, alg[name]), "voxels_per_launch": u}
        elif name == "waves":
            b = lay = WAVE_VISIT_BYTES * visits_instr
            what = {"alg_bytes_per_visit": WAVE_VISIT_BYTES, "visits_per_launch": round(visits_instr, 1)}
        elif name in ("ray_free", "ray_register") and ray_cells:
            b = lay = RAY_CELL_BYTES * ray_cells
            what = {"alg_bytes_per_cell": RAY_CELL_BYTES, "cells_per_launch": ray_cells}
        else:
            return None
        # achieved / frac: what the kernel PHYSICALLY moved (rocprofv3 PMC, the committed profile of this workload on these kernel
        # sources) / its duration in THIS run -- the device's view, never above 1.  Without such a profile: the bytes this build's
        # layout has to move at least (a lower bound of the traffic).  The reference-layout figure of SURVEY 8(d) is kept beside it
        # as *_algorithmic: a kernel that fuses stages or skips what nobody reads moves fewer bytes than the reference's field
        # widths add up to, so that one can exceed the peak and says how much work was avoided, not how busy the memory system is.
        tb = (traffic or {}).get("kernels", {}).get(name)
        phys = tb if tb else lay
        o = {"kernel": name, "avg_launch_ms": round(avg_ms, 4),
             "achieved": round(phys / sec / 1e9, 1), "frac": round(phys / sec / 1e9 / HBM_PEAK_GBS, 4),
             "frac_basis": "pmc" if tb else "layout_lower_bound", "traffic": int(tb) if tb else None,
             "achieved_algorithmic": round(b / sec / 1e9, 1), "frac_algorithmic": round(b / sec / 1e9 / HBM_PEAK_GBS, 4), "alg_bytes_per_launch": int(b),
             "layout_bytes_per_launch": int(lay)}
        o.update(what)
        if o["frac"] > 1.0:                  # not clamped: a fraction above 1 says the byte model is wrong for this launch
            model_errors.append(name)
        return o

    traffic = load_traffic(workload, size, world)
    traffic_note = (traffic or {}).get("stale")
    if traffic_note:
        traffic = None
    sweeps = {k: kernel_roof(k) for k in prof}
    sweeps = {k: v for k, v in sweeps.items() if v}
    # The dominant KERNEL.  The stage timers bracket launches, and three stages are several launches (STAGE_LAUNCHES): a stage's
    # longest kernel lasts at least stage / launches, so a stage only counts as "the dominant kernel" when that lower bound beats every
    # other candidate -- otherwise the two launches of `waves` (0.19 + 0.08 ms on the headline, profiles/r06_c5_kernel_stats.txt)
    # would pass for one kernel longer than pass X's 0.26.  Where a several-launch stage does win (the flood frames of the projective
    # lidar workloads: waves A / B alone are most of the update) the entry is the stage's: its bytes over its time.
    dom = dominant_stage(prof)
    dom_roof = sweeps.get(dom) or {}
    roofline = {"bound": "hbm", "kernel": dom, "achieved": dom_roof.get("achieved"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom_roof.get("frac"), "traffic": dom_roof.get("traffic"),
                "traffic_source": (traffic or {}).get("source") or traffic_note, "csrc_hash": csrc_hash(),
                "avg_launch_ms": round(prof[dom][0] / prof[dom][1], 4)}
    roofline.update({k: v for k, v in dom_roof.items() if k not in roofline and k != "kernel"})
    roofline["note"] = ("dominant kernel of the map update (a stage of several launches only when stage time / launches beats every single kernel).  achieved = traffic / avg_launch_ms, frac = achieved / peak: traffic = HBM bytes per launch "
                        "from the rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate passes; profiles/, named in traffic_source) of this "
                        "workload on THESE kernel sources (csrc_hash; a profile of other sources is withheld and frac falls back to the bytes this "
                        "build's layout must move at least, frac_basis says which); avg_launch_ms from HIP events on the kernel's own dispatch in "
                        "this run (the mapper's stream).  achieved_algorithmic / frac_algorithmic = SURVEY 8(d)'s reference-layout bytes (Mark 33 B + "
                        "commit 37 B per voxel for the fused sweep) x the units of one launch / the same duration: it exceeds the peak when the kernel "
                        "moves fewer bytes than the reference's layout implies.  This device streams reads at 6.5 TB/s and writes at 4.1 TB/s, one "
                        "after the other (tools/sweep_probe.hip): a sweep that mostly writes is at its floor near frac 0.5")

    def group_roof(names, total_ms):
        """several kernels as one sweep: bytes added up, over `total_ms`"""
        ks = [sweeps[k] for k in names if k in sweeps]
        sec = total_ms * 1e-3
        if not ks or sec <= 0:
            return None
        allpmc = all(k["traffic"] for k in ks)
        phys = sum((k["traffic"] if allpmc else k["layout_bytes_per_launch"]) for k in ks)
        ab = sum(k["alg_bytes_per_launch"] for k in ks)
        return {"kernels": [k["kernel"] for k in ks], "ms_per_step": round(total_ms, 4), "achieved": round(phys / sec / 1e9, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(phys / sec / 1e9 / HBM_PEAK_GBS, 4), "frac_basis": "pmc" if allpmc else "layout_lower_bound",
                "traffic": int(phys) if allpmc else None, "alg_bytes_per_step": int(ab),
                "achieved_algorithmic": round(ab / sec / 1e9, 1), "frac_algorithmic": round(ab / sec / 1e9 / HBM_PEAK_GBS, 4)}

    # the wavefront sweep the north-star target names: Mark + obtainFrontiers + waves A/B/C + commit
    ws_ms = sum(prof[k][0] / prof[k][1] for k in WAVEFRONT_SWEEP if k in prof)
    wavefront = group_roof(WAVEFRONT_SWEEP, ws_ms) or {}
    wavefront["per_kernel_ms"] = {k: sweeps[k]["avg_launch_ms"] for k in WAVEFRONT_SWEEP if k in sweeps}
    wavefront["per_kernel_frac"] = {k: sweeps[k]["frac"] for k in WAVEFRONT_SWEEP if k in sweeps}
    update = group_roof(list(sweeps), ms_per_step) or {}
    update["note"] = "every stage of the map update: bytes of all kernels of a step / the uninstrumented step time"

    tgrid = tiling.tile_grid(world)
    out = {
        "value": round(value, 2), "ms_per_step": round(ms_per_step, 4), "hz": round(hz, 3),
        "timed_regions": len(regions), "region_ms": [round(1e3 * x, 3) for x in regions], "timed_s": round(sum(regions), 3),
        "step_ms": ({"median": round(percentile(lat, 0.5), 4), "p95": round(percentile(lat, 0.95), 4), "min": round(min(lat), 4),
                     "max": round(max(lat), 4), "n": len(lat), "how": "one event pair per step on the mapper's stream, separate pass"} if lat else None),
        "config": {"workload": "%dx%dx%d @ %.2f m, %s, full map update, cutoff %.0f m, fast_mode %s"
                               % (size[0], size[1], size[2], voxel, feed2.describe(), cutoff_dist, "on" if fast_mode else "off"),
                   "preset": workload, "baseline_config": BASELINE_CONFIG.get(workload),
                   "grid": list(size), "voxel_m": voxel, "cutoff_m": cutoff_dist, "fast_mode": bool(fast_mode),
                   "drive": ({"mode": DRIVE["mode"], "turn_frames": C5_TURN if DRIVE["mode"] == "turn" else None, "delta_vox": C5["delta_vox"],
                              "retain_radius_blocks": DRIVE["retain"]} if workload == "c5" else
                             {"mode": "turn", "turn_frames": LIDAR_TURN, "delta_vox": PRESETS[workload]["delta"], "retain_radius_blocks": DRIVE["retain"]}),
                   "tiles": ("%dx%dx%d tiles of %dx%dx%d, one per GPU" % (tgrid + tuple(size))) if world > 1 else "single volume",
                   "known_voxel_fraction": round(res["known"], 4),
                   "wave_visits_per_step": [round(visits[k], 1) for k in "abc"],
                   "wave_levels_last_step": [st1["levels_a"], st1["levels_b"], st1["levels_c"]],
                   "blocks": blocks},
        "steps": K, "warmup": W,
        "kernels_ms_per_step": {k: round(v[0] / K, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
        "ms_per_step_instrumented": round(1e3 * dt_instr / K, 4),
        "roofline": roofline, "roofline_wavefront_sweep": wavefront, "roofline_update": update,
        "roofline_sweeps": {k: {kk: vv for kk, vv in v.items() if kk != "kernel"} for k, v in sweeps.items()},
    }
    if world > 1:       # the halo exchange this run timed (VERDICT r4 "next" #3 / #10)
        out["config"].update({
            "exchange": "RCCL" if backend == "nccl" else "gloo, host staging", "rccl_ranks": world if backend == "nccl" else 0,
            "halo_mode": halo_mode, "halo_max_rounds": r.halo_rounds, "rounds_per_update": round(rounds_per_step, 3),
            "updates_unconverged": (round_stats or {}).get("updates_unconverged"), "exchange_notes": notes,
            # what follows the tile's own sweep in a step (face export, transfers, import, merge end, refinement rounds), median over the
            # latency pass; nothing of it overlaps the tile's own work (obtainFrontiers reads the ghosts): overlap_frac 0
            "exchange_ms_per_step": round(percentile(getattr(r, "exchange_ms", []) or [0.0], 0.5), 4), "overlap_frac": 0.0})
    if model_errors:
        out["roofline_model_errors"] = sorted(model_errors)
    if accuracy is not None:
        out["accuracy"] = accuracy
    return out


def accuracy_check(m, voxel):
    """Gnd_truth_checker::cmp_dist (gt_checker.h:30-80) on the local volume: for every known voxel the error between the distance
    to the nearest OCCUPIED voxel of the volume and the published EDT, in metres: RMSE, maximum, and how many voxels are more than
    a millimetre below / above.  The nearest occupied voxel comes from the exact multi-threaded CPU EDT (oracle/edt_mt.c — a checker,
    outside every timed region) instead of the reference's PCL KD-tree; same quantity.  Like the reference's check it only knows the
    obstacles INSIDE the volume: a voxel whose closest obstacle the volume has left behind shows up as "EDT is less"."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    loc = m.read_local(dist_sq=False, coc=False)
    ty, edt = loc["type"], loc["edt"]
    d2, _ = oracle_py.edt_mt(ty, want_coc=False)
    known = ty != 0
    if not (ty == 2).any() or not known.any():
        return {"rms_m": None, "note": "no occupied or no known voxel in the volume"}
    err = (np.sqrt(d2[known].astype(np.float64)) - edt[known].astype(np.float64)) * voxel
    return {"rms_m": round(float(np.sqrt((err ** 2).mean())), 6), "max_err_m": round(float(np.abs(err).max()), 6),
            "edt_less": int((err > 0.001).sum()), "edt_more": int((err < -0.001).sum()), "voxels": int(known.sum()),
            "how": "Gnd_truth_checker::cmp_dist (gt_checker.h:30-80) against the exact CPU EDT of the volume's own obstacles, after the last timed update"}


def costmap_bench(torch, gie, scenes, dev, local_rank, K=10):
    """Row a17 on the clock: the CostMap.msg payload (SeenDist, 8 B/voxel: LocMap::convertCostMap +
    copy_*_2_host, local_batch.h:370-391) a for_motion_planner node publishes after every update.  BASELINE config 2's volume
    (256^3: 134 MB per frame), frames of its depth camera: update + publish per frame through the blocking reader
    (gie_read_costmap into pageable memory, the reference's way) and through gie_costmap_publish / _acquire (pinned memory, copy
    stream: the copy of frame k overlaps update k + 1); and the payload alone at the headline's 512^3 (1.07 GB)."""
    out = {}
    p = PRESETS["c2_projective"]
    cfg = gie.make_config(p["voxel"], p["size"], cutoff_dist=p["cutoff"], fast_mode=p["fast"], device_id=local_rank, for_motion_planner=True,
                          wave_workgroups=WAVE_GRID["wgs"])
    feed = make_feed("c2_projective", torch, scenes, dev, p["voxel"], p["size"], (0, 0, 0), 3 + 3 * K)
    feed.prepare(0, 3 + 3 * K)
    m = gie.Mapper(cfg)
    n = p["size"][0] * p["size"][1] * p["size"][2]
    try:
        for i in range(3):
            feed.step_input(m, i); m.step()
        m.read_costmap(); m.costmap_publish(); m.costmap_acquire(copy=False)          # first use: staging and pinned buffers
        m.sync()
        i = 3
        t0 = time.perf_counter()
        for _ in range(K):
            feed.step_input(m, i); m.step(); i += 1
        m.sync()
        t_upd = (time.perf_counter() - t0) / K
        t0 = time.perf_counter()
        for _ in range(K):
            feed.step_input(m, i); m.step(); m.read_costmap(); i += 1
        t_block = (time.perf_counter() - t0) / K
        t0 = time.perf_counter()
        for _ in range(K):
            feed.step_input(m, i); m.step(); i += 1
            m.costmap_acquire(copy=False)                 # frame k - 1's payload is in host memory by now (or we wait for it)
            m.costmap_publish()
        m.costmap_acquire(copy=False)
        t_async = (time.perf_counter() - t0) / K
        t0 = time.perf_counter()
        for _ in range(3):
            m.costmap_publish(); m.costmap_acquire(copy=False)
        t_pub = (time.perf_counter() - t0) / 3
    finally:
        m.close()
    out["c2_256"] = {"payload_mb": round(n * 8 / 1e6, 1), "update_ms": round(1e3 * t_upd, 3), "update_plus_blocking_read_ms": round(1e3 * t_block, 3),
                     "update_plus_async_publish_ms": round(1e3 * t_async, 3), "publish_alone_ms": round(1e3 * t_pub, 3),
                     "publish_gbps": round(n * 8 / t_pub / 1e9, 1), "frames": K}
    size = (512, 512, 512)
    m = gie.Mapper(gie.make_config(0.05, size, cutoff_dist=2.0, device_id=local_rank, for_motion_planner=True, wave_workgroups=WAVE_GRID["wgs"],
                                   max_blocks=pool_blocks("c5", size, 8)))
    n = size[0] * size[1] * size[2]
    try:
        hw = HashWorldFeed(torch, scenes, dev, 0.05, size, (0, 0, 0))
        hw.prepare(0, 2)
        for i in range(2):
            hw.step_input(m, i); m.step()
        m.costmap_publish(); m.costmap_acquire(copy=False); m.sync()
        t0 = time.perf_counter()
        for _ in range(2):
            m.costmap_publish(); m.costmap_acquire(copy=False)
        t_pub = (time.perf_counter() - t0) / 2
        t0 = time.perf_counter()
        m.read_costmap()
        t_block = time.perf_counter() - t0
    finally:
        m.close()
    out["c5_512"] = {"payload_mb": round(n * 8 / 1e6, 1), "publish_alone_ms": round(1e3 * t_pub, 2), "publish_gbps": round(n * 8 / t_pub / 1e9, 1),
                     "blocking_read_ms": round(1e3 * t_block, 2)}
    return out


TRAFFIC_FILE = "traffic_r06.json"


def csrc_hash():
    """Content hash of the library's sources (csrc/*.h, *.hip, include/*.h): what a committed PMC profile is stamped with, so
    that a profile of OTHER kernels is never priced against this build's durations."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for d, exts in ((os.path.join(ROOT, "gie-mapping_amd", "csrc"), (".h", ".hip")), (os.path.join(ROOT, "include"), (".h",))):
        files += [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith(exts)]
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_traffic(workload, size, world):
    """PMC bytes per map update and stage from the committed rocprofv3 profile of this exact workload (profiles/traffic_r06.json,
    written by tools/profile_round.sh + tools/merge_traffic.py), or {"stale": why} when the profile was taken on other kernel
    sources than this build's, or None."""
    tp = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
    if world != 1 or not os.path.exists(tp):
        return None
    try:
        tj = json.load(open(tp))
    except Exception:
        return None
    e = tj.get("%s_%dx%dx%d" % (workload, size[0], size[1], size[2]))
    if not e:
        return None
    here = csrc_hash()
    if e.get("csrc_hash") != here:
        return {"stale": "profiles/%s was taken on kernel sources %s, this build is %s: traffic withheld" % (TRAFFIC_FILE, e.get("csrc_hash"), here)}
    return e


def cpu_baseline(scenes, torch, dev, voxel, size, cutoff_dist, workload, W):
    """(i) the timed full-size CPU baseline of BASELINE.md §2 row B1: exact separable EDT with closest-obstacle tracking on
    the SAME local grid, multi-threaded over all host cores (oracle/edt_mt.c, validated against brute force in tests/);
    (ii) the scalar CPU restatement of the whole map update (the oracle) on one core, on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    import gie
    cores = os.cpu_count() or 1
    # the occupancy the EDT works on: frame W of the workload (for the hash world the fused types are the labels: a first
    # observation of an obstacle gives 250 * 0.8 = 200 > 180)
    if PRESETS[workload]["feed"] == "hash":
        pos, _ = c5_pose(scenes, W, voxel)
        pvt = scenes.local_pivot(pos, voxel, size)
        lab = scenes.hash_world_labels(pvt, size, W, seed=C5["seed"], p_occ=C5["p_occ"], toggle_frac=C5["toggle_frac"],
                                       arange=lambda n: torch.arange(n, dtype=torch.int64, device=dev),
                                       where=lambda c, a, b: torch.where(c, torch.tensor(a, dtype=torch.int8, device=dev), torch.tensor(b, dtype=torch.int8, device=dev)))
        types = lab.cpu().numpy()
        del lab
    else:
        feed = make_feed(workload, torch, scenes, dev, voxel, size, (0, 0, 0), W + 1)
        cfg = gie.make_config(voxel, size, cutoff_dist=cutoff_dist, fast_mode=PRESETS[workload]["fast"], device_id=dev.index or 0)
        m = gie.Mapper(cfg)
        feed.prepare(0, W + 1)
        for i in range(W + 1):
            feed.step_input(m, i)
            m.step()
        types = m.read_local(edt=False, dist_sq=False, coc=False)["type"]
        m.close()
    n = size[0] * size[1] * size[2]
    t0 = time.perf_counter()
    oracle_py.edt_mt(types, nthreads=cores)
    t1 = time.perf_counter() - t0
    reps = max(1, min(8, int(12.0 / max(t1, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle_py.edt_mt(types, nthreads=cores)
    dt = (time.perf_counter() - t0) / reps
    out = {"value": round(n / dt / 1e6, 2), "unit": "Mvoxels/s", "cores": cores, "host_cores": cores, "kind": "port",
           "stage": "batch EDT only (exact separable EDT + closest obstacle), %d threads" % cores,
           "ms_per_update": round(1e3 * dt, 2),
           "sample": "%dx%dx%d grid of frame %d (%d obstacles), %d+1 repetitions, %.1f s"
                     % (size[0], size[1], size[2], W, int((types == 2).sum()), reps, t1 + dt * reps)}
    del types
    # the whole map update, scalar port on one core, bounded sample (128^3 under full observation, 256^3 for the sparse lidar scans)
    hashw = PRESETS[workload]["feed"] == "hash"
    s2 = (128, 128, 128) if (hashw or workload.endswith("_projective")) else tuple(min(256, e) for e in size)
    cfg = gie.make_config(voxel, s2, cutoff_dist=min(cutoff_dist, 5.0), fast_mode=PRESETS[workload]["fast"])
    om = oracle_py.OracleMapper(cfg)
    lf = None if hashw else make_feed(workload, torch, scenes, torch.device("cpu"), voxel, s2, (0, 0, 0), 32)
    t0 = time.perf_counter()
    k = 0
    while k < 32 and (k < 3 or time.perf_counter() - t0 < 10.0):
        if hashw:
            pos, q = c5_pose(scenes, k, voxel)
            lab = scenes.hash_world_labels(scenes.local_pivot(pos, voxel, s2), s2, k, seed=C5["seed"], p_occ=C5["p_occ"], toggle_frac=C5["toggle_frac"])
            om.update(pos, q, "labels", lab.astype(np.int8))
        else:
            lf.oracle_update(om, k)
        k += 1
    dt2 = time.perf_counter() - t0
    om.close()
    out["full_update_1core"] = {"value": round(s2[0] * s2[1] * s2[2] * k / dt2 / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
                                "stage": "whole map update, scalar restatement (oracle/gie_oracle.c)",
                                "sample": "%dx%dx%d grid, same generator, %d map updates, %.1f s incl. input generation" % (s2[0], s2[1], s2[2], k, dt2)}
    return out


WAVE_GRID = {"wgs": int(os.environ.get("GIE_BENCH_WAVE_WGS", "160"))}     # gie_config.wave_workgroups: a rank owns its device (160 of 256 measured best); ranks sharing a device: 192 / ranks
PLACE_TRIES = 4              # gie_config.place_tries: gie_create re-draws the sweep's planes against a probe.  Round 4: 0.76 vs 0.84 ms of Mark + commit by placement;
                             # round 5 (one large non-temporal write stream): 0.471 - 0.474 ms placed against 0.472 - 0.497 as allocated (tools/place_variance.py, A/B on one box)
PARTIAL = {}     # per workload: the regions timed so far (printed with an "error" key if the run dies later on)
FULL_FILE = os.path.join("profiles", "bench_last_full.json")
LINE_LIMIT = 8000           # bytes of the ONE line on stdout (a 21.8 KB line left the driver's record unparsed)


def _short(v, n=120):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "~"


def compact_roofline(r):
    """The ONE roofline object of the line: the contract's keys + what prices it."""
    keys = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "frac_basis", "frac_algorithmic",
            "achieved_algorithmic", "alg_bytes_per_launch", "layout_bytes_per_launch", "voxels_per_launch", "csrc_hash")
    return {k: r.get(k) for k in keys if k in r}


def build_line(main_res, extras, cpu, n_gpus, metric="edt_map_update_throughput", full_file=FULL_FILE, costmap=None):
    """(line, full): `line` is what rank 0 prints — headline keys, ONE roofline (dominant kernel), cpu_baseline, per-workload step
    times and fractions, below LINE_LIMIT bytes, no string above 120 characters; `full` is everything (written to full_file)."""
    full = {"metric": metric, "value": main_res["value"], "unit": "Mvoxels/s", "n_gpus": n_gpus, "steps": main_res["steps"], "warmup": main_res["warmup"],
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic"}
    full.update({k: v for k, v in main_res.items() if k not in full})
    full["extra_runs"] = extras
    if cpu is not None:
        full["cpu_baseline"] = cpu
    if costmap is not None:
        full["costmap_publish"] = costmap
    full["notes"] = {
        "timing": "value / ms_per_step: median of the timed regions, each EXACTLY K steps between barrier + synchronize, MAX over ranks; kernels_ms_per_step and "
                  "the roofline objects: the first region replayed on a fresh mapper with start / stop events on every kernel's dispatch",
        "roofline": "dominant kernel of the map update.  achieved = traffic / avg_launch_ms, frac = achieved / peak: traffic = HBM bytes per launch from the rocprofv3 "
                    "PMC passes (separate passes; profiles/) of this workload on THESE kernel sources (csrc_hash; a profile of other sources is withheld and frac "
                    "falls back to the bytes this build's layout must move at least: frac_basis); avg_launch_ms from HIP events on the kernel's own dispatch in this "
                    "run.  *_algorithmic = SURVEY 8(d)'s reference-layout bytes x the units of one launch / the same duration; it exceeds 1 when the kernel moves "
                    "fewer bytes than the reference's layout implies.  Not clamped: a frac above 1 is listed under roofline_model_errors",
    }
    cfgm = main_res["config"]
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["hz"] = main_res["hz"]
    line["config"] = {k: _short(v) for k, v in cfgm.items() if k not in ("drive", "exchange_notes", "wave_levels_last_step")}
    line["roofline"] = compact_roofline(main_res["roofline"])
    ws, up = main_res.get("roofline_wavefront_sweep") or {}, main_res.get("roofline_update") or {}
    line["roofline_wavefront_sweep"] = {k: ws.get(k) for k in ("ms_per_step", "frac", "frac_basis", "frac_algorithmic") if k in ws}
    line["roofline_update"] = {k: up.get(k) for k in ("ms_per_step", "frac", "frac_basis", "frac_algorithmic") if k in up}
    if main_res.get("step_ms"):
        line["step_ms"] = {k: main_res["step_ms"][k] for k in ("median", "p95")}
    line["timed_regions"], line["timed_s"] = main_res["timed_regions"], main_res["timed_s"]
    line["kernels_ms_per_step"] = main_res["kernels_ms_per_step"]
    if main_res.get("roofline_model_errors"):
        line["roofline_model_errors"] = main_res["roofline_model_errors"]
    if cpu is not None:
        line["cpu_baseline"] = {k: (_short(v) if not isinstance(v, dict) else {kk: _short(vv) for kk, vv in v.items()}) for k, v in cpu.items()}
        line["config"]["cpu_baseline_stage"] = _short(cpu.get("stage", ""))
    if extras:
        preset = cfgm.get("preset", "main")
        line["ms_per_step_by_workload"] = {preset: main_res["ms_per_step"], **{wl: e["ms_per_step"] for wl, e in extras.items()}}
        line["mvoxels_per_s_by_workload"] = {preset: main_res["value"], **{wl: e["value"] for wl, e in extras.items()}}
        # one fraction each: the dominant kernel of that workload's update (name : frac, physical; frac_basis in the full file)
        line["frac_by_workload"] = {preset: [main_res["roofline"]["kernel"], main_res["roofline"]["frac"]],
                                    **{wl: [e["roofline"]["kernel"], e["roofline"]["frac"]] for wl, e in extras.items()}}
        line["baseline_config_by_workload"] = {wl: BASELINE_CONFIG[wl] for wl in [preset] + list(extras) if wl in BASELINE_CONFIG}
        ogm = {}
        for wl, kern in (("vlp16_projective", "ogm_classify"), ("vlp16", "ray_free"), ("c2", "ray_free"), ("c2_projective", "ogm_classify")):
            rr = (extras.get(wl) or {}).get("roofline_sweeps", {}).get(kern)
            if rr:
                ogm["%s:%s" % (wl, kern)] = {k: rr.get(k) for k in ("avg_launch_ms", "frac", "frac_basis")}
        if ogm:
            line["roofline_ogm"] = ogm
    if main_res.get("accuracy") is not None:
        line["accuracy"] = {k: _short(v) for k, v in main_res["accuracy"].items()}
    if costmap is not None:          # row a17 (CostMap.msg payload, 8 B/voxel), off the timed region: never part of `value`
        line["costmap_publish"] = costmap
    line["detail"] = full_file
    # the limit holds whatever a run produces: drop the least important blocks first
    for k in ("roofline_ogm", "costmap_publish", "kernels_ms_per_step", "mvoxels_per_s_by_workload", "baseline_config_by_workload", "accuracy", "roofline_update", "step_ms"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(k, None)
    return line, full


def write_full(full, path=FULL_FILE):
    """Everything the line leaves out, beside the committed profiles (and under gpurun_out/ when that exists: it travels back from a GPU box)."""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if d != ROOT and not os.path.isdir(d):
                continue
            f = os.path.join(d, path) if d == ROOT else os.path.join(d, os.path.basename(path))
            os.makedirs(os.path.dirname(f), exist_ok=True)
            with open(f, "w") as fh:
                json.dump(full, fh, indent=1)
        except OSError as e:                 # a read-only checkout must not cost the line
            sys.stderr.write("bench: could not write %s: %s\n" % (path, e))


def main():
    rank = int(os.environ.get("RANK", "0"))
    try:
        run_bench()
    except BaseException as e:          # noqa: B902 — the ONE JSON line is printed whatever happens, with what was timed
        if isinstance(e, SystemExit) and e.code in (0, None):
            raise
        if rank == 0:
            line = {"metric": "edt_map_update_throughput", "value": None, "unit": "Mvoxels/s", "higher_is_better": True,
                    "error": _short("%s: %s" % (type(e).__name__, e), 400), "partial": {}}
            for wl, p in PARTIAL.items():
                rs = sorted(p["region_s"])
                if rs:
                    med = rs[len(rs) // 2]
                    line["partial"][wl] = {"timed_regions": len(rs), "ms_per_step": round(1e3 * med / p["steps"], 4),
                                           "value": round(p["n_voxels"] * p["steps"] / med / 1e6, 2)}
            print(json.dumps(line), flush=True)
        raise


def run_bench():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, nargs=3, default=None, help="local grid (default: the workload's preset)")
    ap.add_argument("--voxel", type=float, default=None, help="voxel width in m (default: the preset's)")
    ap.add_argument("--cutoff", type=float, default=None, help="wave cutoff distance in m (default: the preset's)")
    ap.add_argument("--workload", "--sensor", dest="workload", choices=WORKLOADS, default="c5")
    ap.add_argument("--min-timed-s", type=float, default=0.5, help="repeat the K-step region until this much has been timed")
    ap.add_argument("--drive", choices=["turn", "line"], default="turn", help="c5 robot: 24 frames out and 24 back (default), or a straight line forever")
    ap.add_argument("--retain", type=int, default=0, help="gie_config.retain_radius_blocks: erase and recycle blocks more than R blocks outside the volume (0: never, the reference's rule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rms", action="store_true",
                    help="accuracy profiler (the reference's Gnd_truth_checker, gt_checker.h:30-80): RMSE of the local EDT after the last "
                         "timed update against the exact distance to the nearest occupied voxel of the volume")
    ap.add_argument("--no-extras", "--no-secondary", dest="no_extras", action="store_true",
                    help="skip the runs reported beside the headline (lidar workloads, BASELINE configs 2 / 3 / 4)")
    ap.add_argument("--extras", default=None, help="comma list of workloads to run beside the headline (default: %s)" % ",".join(DEFAULT_EXTRAS))
    args = ap.parse_args()
    DRIVE["mode"], DRIVE["retain"] = args.drive, max(0, args.retain)

    import torch
    import gie
    from gie import scenes, tiling

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
        # the wavefront kernel is persistent and takes a whole compute unit's LDS per workgroup (wave C's tiles): the grids of
        # all ranks have to be resident side by side on the one device, or their grid barriers wait for each other until they time out
        WAVE_GRID["wgs"] = max(8, 192 // max(1, world))
    torch.cuda.set_device(local_rank)
    dist = None
    group, transport_note = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane on gloo; the face layers over RCCL if its pre-flight passes on every rank, else staged through the host
        # (gloo announces its connections on the process's stdout, from C++: the ONE line of this program is the only thing that
        #  belongs there, so file descriptor 1 points at stderr while the groups are set up)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            tr = tiling.init_transport(torch, dist, rank, world, torch.device("cuda", local_rank), want=backend)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        backend, group, transport_note = tr["backend"], tr["group"], tr["note"]
        if transport_note and rank == 0:
            sys.stderr.write("bench: %s\n" % transport_note)

    dev = torch.device("cuda", local_rank)
    W, K = args.warmup, args.steps

    def run(wl, **kw):
        p = PRESETS[wl]
        size = tuple(args.size) if (args.size and wl == args.workload) else p["size"]
        voxel = args.voxel if (args.voxel and wl == args.workload) else p["voxel"]
        cutoff = args.cutoff if (args.cutoff and wl == args.workload) else p["cutoff"]
        return run_workload(torch, gie, scenes, tiling, kw.pop("dist", None), wl, size, voxel, cutoff, kw.pop("W", W), kw.pop("K", K), kw.pop("rank", 0),
                            kw.pop("world", 1), dev, local_rank, backend, fast_mode=p["fast"], **kw), (size, voxel, cutoff)

    main_res, (size, voxel, cutoff) = run(args.workload, dist=dist, rank=rank, world=world, min_timed_s=args.min_timed_s, rms=args.rms, group=group,
                                          transport_note=transport_note)
    if rank == 0:
        extras = {}
        if world == 1 and not args.no_extras:
            names = [w for w in (args.extras.split(",") if args.extras else DEFAULT_EXTRAS) if w and w != args.workload]
            for wl in names:
                # short runs: W / K bounded (the synthetic scans of a workload are generated once, outside every timed region)
                e, _ = run(wl, W=min(W, 3), K=min(K, 10), min_timed_s=0.1, max_regions=3, with_latency=False)
                extras[wl] = e
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(scenes, torch, dev, voxel, size, cutoff, args.workload, W)
        costmap = None
        if world == 1 and not args.no_extras:
            costmap = costmap_bench(torch, gie, scenes, dev, local_rank)
        line, full = build_line(main_res, extras, cpu, world, costmap=costmap)
        if world == 1 and not args.no_extras:
            # the headline workload once more as a maintainer following INTEGRATION.md gets it: gie_config's defaults (wave grid = half the
            # compute units so that two mappers share a device, no plane placement at gie_create) instead of this program's knobs
            dk, _ = run(args.workload, W=min(W, 3), K=min(K, 10), min_timed_s=0.1, max_regions=3, with_latency=False, default_knobs=True)
            line["ms_per_step_library_defaults"] = dk["ms_per_step"]
            full["library_defaults_run"] = {"ms_per_step": dk["ms_per_step"], "knobs": "wave_workgroups = 0 (CUs / 2), place_tries = 0",
                                            "headline_knobs": "wave_workgroups = %d, place_tries = %d" % (WAVE_GRID["wgs"], PLACE_TRIES)}
        write_full(full)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()                                    # rank 0's instrumented pass is over
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
