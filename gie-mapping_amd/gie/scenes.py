"""Seeded synthetic sensor frames: a world of axis-aligned boxes over a ground
slab, a fraction of which toggles present/absent every frame so that all three wavefronts are
exercised, seen by a moving pinhole depth camera or a multi-ring lidar.

Pure numpy, used by tests/ and bench.py to produce INPUT data only (depth images, range images,
point clouds, poses).  Sensor-frame convention is the reference's: x forward, y left, z up;
depth = x (src/kernel/realsense/camera_helper.h:11-38).
"""
import math

import numpy as np


class BoxWorld:
    def __init__(self, seed, extent, n_boxes=48, toggle_frac=0.25, ground_z=-1.0, min_size=0.3, max_size=1.6):
        rng = np.random.default_rng(seed)
        ext = np.asarray(extent, dtype=np.float64)  # half-extent of the populated region (m)
        ctr = rng.uniform(-ext, ext, size=(n_boxes, 3))
        ctr[:, 2] = rng.uniform(ground_z, ground_z + 2.0 * 1.2, size=n_boxes)
        half = rng.uniform(min_size, max_size, size=(n_boxes, 3)) * 0.5
        lo = ctr - half
        hi = ctr + half
        # keep the sensor start position (origin) free
        near = np.all((lo < 0.6) & (hi > -0.6), axis=1)
        lo[near] += 2.0
        hi[near] += 2.0
        ground_lo = np.array([[-4.0 * ext[0], -4.0 * ext[1], ground_z - 0.3]])
        ground_hi = np.array([[4.0 * ext[0], 4.0 * ext[1], ground_z]])
        self.lo = np.vstack([ground_lo, lo])
        self.hi = np.vstack([ground_hi, hi])
        self.toggles = np.zeros(n_boxes + 1, dtype=bool)
        k = int(round(toggle_frac * n_boxes))
        if k > 0:
            self.toggles[1 + rng.choice(n_boxes, size=k, replace=False)] = True
        self.phase = rng.integers(0, 2, size=n_boxes + 1)

    def active(self, frame):
        on = np.ones(self.lo.shape[0], dtype=bool)
        t = self.toggles
        on[t] = ((frame + self.phase[t]) % 2) == 0
        return on

    def cast(self, origin, dirs, frame, chunk=65536, device=None):
        """Nearest hit parameter t >= 0 along origin + t*dirs (dirs: [R,3]); inf when none.
        device: a torch device to run the slab test on (bench.py: the GPU; same float64 operations in the same order,
        so the result is the numpy one) -- None = numpy."""
        on = self.active(frame)
        lo = self.lo[on]
        hi = self.hi[on]
        o = np.asarray(origin, dtype=np.float64)
        if device is not None:
            return self._cast_torch(o, lo, hi, dirs, chunk, device)
        out = np.full(dirs.shape[0], np.inf)
        for s in range(0, dirs.shape[0], chunk):
            d = dirs[s:s + chunk].astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = 1.0 / d
                t0 = (lo[None, :, :] - o[None, None, :]) * inv[:, None, :]
                t1 = (hi[None, :, :] - o[None, None, :]) * inv[:, None, :]
            tmin = np.minimum(t0, t1)
            tmax = np.maximum(t0, t1)
            # rays parallel to a slab: inside → (-inf, inf), outside → no hit
            par = d[:, None, :] == 0.0
            inside = (o[None, None, :] >= lo[None, :, :]) & (o[None, None, :] <= hi[None, :, :])
            tmin = np.where(par, np.where(inside, -np.inf, np.inf), tmin)
            tmax = np.where(par, np.where(inside, np.inf, -np.inf), tmax)
            tn = tmin.max(axis=2)
            tf = tmax.min(axis=2)
            hit = (tf >= tn) & (tf >= 0.0)
            tn = np.where(hit, np.maximum(tn, 0.0), np.inf)
            out[s:s + chunk] = tn.min(axis=1)
        return out

    @staticmethod
    def _cast_torch(o, lo, hi, dirs, chunk, device):
        """cast() with torch tensors on `device`: the numpy statement above, operation for operation, in float64."""
        import torch
        f8 = torch.float64
        lo_t = torch.from_numpy(np.ascontiguousarray(lo)).to(device)[None, :, :]
        hi_t = torch.from_numpy(np.ascontiguousarray(hi)).to(device)[None, :, :]
        o_t = torch.from_numpy(o).to(device)[None, None, :]
        d_all = torch.from_numpy(np.ascontiguousarray(dirs, dtype=np.float64)).to(device)
        pinf = torch.tensor(float("inf"), dtype=f8, device=device)
        ninf = torch.tensor(float("-inf"), dtype=f8, device=device)
        zero = torch.tensor(0.0, dtype=f8, device=device)
        inside = (o_t >= lo_t) & (o_t <= hi_t)
        out = torch.empty(d_all.shape[0], dtype=f8, device=device)
        for s in range(0, d_all.shape[0], chunk):
            d = d_all[s:s + chunk]
            inv = (1.0 / d)[:, None, :]
            t0 = (lo_t - o_t) * inv
            t1 = (hi_t - o_t) * inv
            tmin = torch.minimum(t0, t1)
            tmax = torch.maximum(t0, t1)
            par = (d == 0.0)[:, None, :]
            tmin = torch.where(par, torch.where(inside, ninf, pinf), tmin)
            tmax = torch.where(par, torch.where(inside, pinf, ninf), tmax)
            tn = tmin.amax(dim=2)
            tf = tmax.amin(dim=2)
            hit = (tf >= tn) & (tf >= 0.0)
            tn = torch.where(hit, torch.maximum(tn, zero), pinf)
            out[s:s + chunk] = tn.amin(dim=1)
        return out.cpu().numpy()


def yaw_quat(yaw):
    return (math.cos(0.5 * yaw), 0.0, 0.0, math.sin(0.5 * yaw))


def rot_from_quat(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def pose(frame, voxel_width, delta_vox=4, yaw_deg=2.0, z=0.0):
    """Sensor pose of `frame` (0-based): +delta_vox voxels in x and yaw_deg of yaw per frame."""
    pos = (np.float32(frame * delta_vox * voxel_width), np.float32(0.0), np.float32(z))
    return pos, yaw_quat(math.radians(yaw_deg) * frame)


def depth_frame(world, frame, pos, quat, rows=480, cols=640, fx=525.0, fy=525.0, cx=319.5, cy=239.5,
                max_depth=8.0, device=None):
    """Pinhole depth image (float32 [rows, cols]); NaN where nothing is hit within max_depth."""
    u, v = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    d_s = np.stack([np.ones_like(u), (cx - u) / fx, (cy - v) / fy], axis=-1).reshape(-1, 3)
    d_w = d_s @ rot_from_quat(quat).T
    t = world.cast(pos, d_w, frame, device=device)
    depth = np.where(t <= max_depth, t, np.nan).astype(np.float32)
    return depth.reshape(rows, cols)


def depth_to_points(depth, fx=525.0, fy=525.0, cx=319.5, cy=239.5):
    """Sensor-frame point cloud of the valid pixels (CAM_HELPER::L2G without the pose)."""
    rows, cols = depth.shape
    u, v = np.meshgrid(np.arange(cols, dtype=np.float32), np.arange(rows, dtype=np.float32))
    ok = np.isfinite(depth)
    d = depth[ok]
    pts = np.stack([d, (np.float32(cx) - u[ok]) * d / np.float32(fx), (np.float32(cy) - v[ok]) * d / np.float32(fy)],
                   axis=-1)
    return np.ascontiguousarray(pts, dtype=np.float32)


def lidar_frame(world, frame, pos, quat, rings=16, az=1800, phi_min_deg=-15.0, phi_inc_deg=2.0, max_range=100.0, device=None):
    """Multi-ring lidar: returns (points_sensor_frame [P,3] float32, hit range per (ring, az))."""
    phi = np.radians(phi_min_deg + phi_inc_deg * np.arange(rings))
    th = -np.pi + 2.0 * np.pi * (np.arange(az) + 0.5) / az
    ph, tt = np.meshgrid(phi, th, indexing="ij")
    d_s = np.stack([np.cos(ph) * np.cos(tt), np.cos(ph) * np.sin(tt), np.sin(ph)], axis=-1).reshape(-1, 3)
    d_w = d_s @ rot_from_quat(quat).T
    t = world.cast(pos, d_w, frame, device=device)
    ok = t <= max_range
    pts = (d_s[ok] * t[ok, None]).astype(np.float32)
    return np.ascontiguousarray(pts), np.where(ok, t, np.inf).reshape(rings, az)


def range_image(points, scan_num=440, ring_num=16, phi_min_deg=-15.0, phi_inc_deg=2.0):
    """Vlp16MapMaker::convertPyntCld binning (src/vlp16_map_maker.cpp:73-147): horizontal range
    per (ring, azimuth bin), +inf where no return.  Ring = nearest elevation ring."""
    img = np.full((ring_num, scan_num), np.inf, dtype=np.float32)
    if points.shape[0] == 0:
        return img
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    res = np.float32(2.0 * math.pi / scan_num)
    b = ((np.arctan2(y, x).astype(np.float32) + np.float32(math.pi)) / res).astype(np.int64)
    hor = np.sqrt(x * x + y * y).astype(np.float32)
    ring = np.rint((np.degrees(np.arctan2(z, hor)) - phi_min_deg) / phi_inc_deg).astype(np.int64)
    ok = (b >= 0) & (b < scan_num) & (ring >= 0) & (ring < ring_num)
    img[ring[ok], b[ok]] = hor[ok]
    return img


# ---- the sensor-less synthetic world of BASELINE config 5 -----------------------
# "voxel (x,y,z) occupied iff hash32(x,y,z,seed,frame_epoch) < p * 2^32, full observation; 25 % of the
# obstacles toggle per frame".  Written with operators only, so the same code runs on numpy int64
# arrays (tests) and torch int64 tensors (bench.py generates the label planes on the GPU).
def _mix32(h):
    h = h & 0xffffffff
    h = h ^ (h >> 16)
    h = (h * 0x7feb352d) & 0xffffffff
    h = h ^ (h >> 15)
    h = (h * 0x846ca68b) & 0xffffffff        # the int64 product may wrap: its low 32 bits do not care
    h = h ^ (h >> 16)
    return h


def pos2coord(p, voxel_width):
    """LocMap::pos2coord (local_batch.h:250-258): floorf(p / w + 0.5f) in fp32."""
    return int(np.floor(np.float32(p) / np.float32(voxel_width) + np.float32(0.5)))


def local_pivot(pos, voxel_width, size, tile_off=(0, 0, 0)):
    """calculate_pivot_origin (local_batch.h:128-142) + the tile offset of gie_set_tile."""
    return tuple(pos2coord(pos[i], voxel_width) - size[i] // 2 + int(tile_off[i]) for i in range(3))


def hash_world_labels(pvt, size, frame, seed=5, p_occ=0.01, toggle_frac=0.25, arange=None, where=None):
    """Label plane [Z][Y][X] (int8: 1 free, 2 occupied) of the local volume with pivot `pvt` at `frame`.
    arange(n) / where(c, a, b) default to numpy; pass torch twins to build it on a device."""
    if arange is None:
        arange = lambda n: np.arange(n, dtype=np.int64)                     # noqa: E731
        where = np.where
    X, Y, Z = size
    gx = (arange(X) + int(pvt[0])).reshape(1, 1, X)
    gy = (arange(Y) + int(pvt[1])).reshape(1, Y, 1)
    gz = (arange(Z) + int(pvt[2])).reshape(Z, 1, 1)
    h = _mix32((gx * 73856093) ^ (gy * 19349669) ^ (gz * 83492791) ^ (int(seed) * 0x9e3779b1))
    base = h < int(p_occ * 4294967296.0)
    h2 = _mix32(h ^ 0x5bd1e995)
    toggler = (h2 >> 8) < int(toggle_frac * 16777216.0)
    present = base & (~toggler | (((h2 & 1) + int(frame)) % 2 == 0))
    return where(present, 2, 1)
