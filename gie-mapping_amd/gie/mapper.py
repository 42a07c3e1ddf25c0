"""Host-side mirror of the reference's per-frame interface, over the C-ABI (include/gie.h).

`Mapper` plays the role of the objects VOLMAPNODE owns (LocMap + the four *MapMaker adapters +
GlbHashMap + the EDT launcher; src/volumetric_mapper.cpp:73-83) and `update()` reproduces the
call order of VOLMAPNODE::publishMap (src/volumetric_mapper.cpp:138-224).  It loads
libgie_hip.so and fails loudly when it is missing — there is no CPU fallback in this package.
"""
import ctypes as C
import math
import os
import sys

import numpy as np

from . import _capi
from ._capi import CamParam, Config, CostMapHdr, FrameStats, MultiScanParam, ScanParam, Voxel

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "csrc", "libgie_hip.so")

SEENDIST_DTYPE = np.dtype([("d", "<f4"), ("s", "u1"), ("o", "u1"), ("pad", "u1", (2,))])
HALO_DTYPE = np.dtype([("dist_sq", "<i4"), ("coc", "<i4", (3,)), ("vox_type", "i1"), ("occ_val", "u1"), ("pad", "i1", (2,))])
HALO_ENTRY_DTYPE = np.dtype([("index", "<i4"), ("v", HALO_DTYPE)])          # gie_halo_entry: a known voxel of a sparse face layer
VOXEL_DTYPE = np.dtype([("occ_val", "u1"), ("vox_type", "i1"), ("pad", "<i2"), ("dist_sq", "<i4"),
                        ("coc", "<i4", (3,))])


def flt2grids_sq(rad, voxel_width):
    """Parameters::flt2GridsSq (include/parameters.h:134-138)."""
    g = int(math.ceil(np.float32(rad) / np.float32(voxel_width)))
    return g * g


def make_config(voxel_width, local_size, occupancy_threshold=180, ogm_min_h=-1000.0, ogm_max_h=1000.0,
                cutoff_dist=None, cutoff_grids_sq=None, fast_mode=False, for_motion_planner=False,
                robot_r=0.4, max_blocks=0, device_id=0, retain_radius_blocks=0, wave_workgroups=0, place_tries=0):
    cfg = Config()
    cfg.voxel_width = voxel_width
    cfg.local_size[:] = [int(v) for v in local_size]
    cfg.occupancy_threshold = occupancy_threshold
    cfg.ogm_min_h = ogm_min_h
    cfg.ogm_max_h = ogm_max_h
    if cutoff_grids_sq is None:
        cutoff_grids_sq = flt2grids_sq(6.0 if cutoff_dist is None else cutoff_dist, voxel_width)
    cfg.cutoff_grids_sq = int(cutoff_grids_sq)
    cfg.fast_mode = int(bool(fast_mode))
    cfg.for_motion_planner = int(bool(for_motion_planner))
    cfg.robot_r2_grids = flt2grids_sq(robot_r, voxel_width)
    cfg.max_blocks = int(max_blocks)
    cfg.device_id = int(device_id)
    cfg.retain_radius_blocks = int(retain_radius_blocks)
    cfg.wave_workgroups = int(wave_workgroups)
    cfg.place_tries = int(place_tries)
    return cfg


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class MapperBase:
    """Everything that only needs a bound function table (see _capi.bind)."""

    def __init__(self, fns, cfg):
        self._f = fns
        self.cfg = cfg
        self.size = tuple(int(v) for v in cfg.local_size)
        self.n = self.size[0] * self.size[1] * self.size[2]
        self._h = fns["create"](C.byref(cfg))
        if not self._h:
            raise RuntimeError("gie_create failed: " + self._err())

    def _err(self):
        f = self._f.get("last_error")
        return f().decode() if f else "error"

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("gie call failed (%d): %s" % (rc, self._err()))

    def close(self):
        if self._h:
            self._f["destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- frame input -----------------------------------------------------------------
    def set_pose(self, pos, quat_wxyz=(1.0, 0.0, 0.0, 0.0)):
        p = (C.c_float * 3)(*[float(v) for v in pos])
        q = (C.c_float * 4)(*[float(v) for v in quat_wxyz])
        self._chk(self._f["set_pose"](self._h, p, q))

    def ogm_pointcloud(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self._chk(self._f["ogm_pointcloud"](self._h, _ptr(xyz), xyz.shape[0]))

    def ogm_multiscan(self, ranges, theta_inc, theta_min, phi_inc, phi_min, max_r=100.0):
        ranges = np.ascontiguousarray(ranges, dtype=np.float32)
        p = MultiScanParam(ranges.shape[1], ranges.shape[0], max_r, theta_inc, theta_min, phi_inc, phi_min)
        self._chk(self._f["ogm_multiscan"](self._h, _ptr(ranges), C.byref(p)))

    def ogm_depth(self, depth, cx, cy, fx, fy, valid_nan=False):
        depth = np.ascontiguousarray(depth, dtype=np.float32)
        p = CamParam(depth.shape[0], depth.shape[1], cx, cy, fx, fy, int(valid_nan))
        self._chk(self._f["ogm_depth"](self._h, _ptr(depth), C.byref(p)))

    def ogm_scan2d(self, ranges, theta_inc, theta_min, max_r=30.0):
        ranges = np.ascontiguousarray(ranges, dtype=np.float32).reshape(-1)
        p = ScanParam(ranges.shape[0], max_r, theta_inc, theta_min)
        self._chk(self._f["ogm_scan2d"](self._h, _ptr(ranges), C.byref(p)))

    def ogm_labels(self, labels):
        """A pre-classified scan: int8 [Z][Y][X], 0 unknown / 1 free / 2 occupied."""
        labels = np.ascontiguousarray(labels, dtype=np.int8)
        assert labels.size == self.n
        self._chk(self._f["ogm_labels"](self._h, _ptr(labels)))

    def set_ext_boxes(self, ll, ur, active):
        ll = np.ascontiguousarray(ll, dtype=np.float32).reshape(-1, 3)
        ur = np.ascontiguousarray(ur, dtype=np.float32).reshape(-1, 3)
        act = np.ascontiguousarray(active, dtype=np.uint8).reshape(-1)
        self._chk(self._f["set_ext_boxes"](self._h, _ptr(ll), _ptr(ur), _ptr(act), ll.shape[0]))

    # --- stages ----------------------------------------------------------------------
    def fuse(self):
        self._chk(self._f["fuse"](self._h))

    def batch_edt(self):
        self._chk(self._f["batch_edt"](self._h))

    def merge(self):
        self._chk(self._f["merge"](self._h))

    def merge_begin_tiled(self):
        self._chk(self._f["merge_begin_tiled"](self._h))

    def merge_end(self):
        self._chk(self._f["merge_end"](self._h))

    def step(self):
        self._chk(self._f["step"](self._h))

    def step_begin_tiled(self):
        """fuse + batch EDT + the first half of the merge of a tiled run (the face layers are exported next)"""
        self.fuse(); self.batch_edt(); self.merge_begin_tiled()

    def sync(self):
        f = self._f.get("sync")
        if f:
            self._chk(f(self._h))

    # --- readers (arrays come back shaped [Z][Y][X], x fastest) ------------------------
    def _shape(self):
        return (self.size[2], self.size[1], self.size[0])

    def read_local(self, edt=True, vtype=True, dist_sq=True, coc=True):
        out = {}
        e = np.empty(self._shape(), np.float32) if edt else None
        t = np.empty(self._shape(), np.int8) if vtype else None
        d = np.empty(self._shape(), np.int32) if dist_sq else None
        c = np.empty(self._shape() + (3,), np.int32) if coc else None
        self._chk(self._f["read_local"](self._h, _ptr(e), _ptr(t), _ptr(d), _ptr(c)))
        for k, v in (("edt", e), ("type", t), ("dist_sq", d), ("coc", c)):
            if v is not None:
                out[k] = v
        return out

    def read_ogm(self):
        t = np.empty(self._shape(), np.int8)
        r = np.empty(self._shape(), np.int32)
        self._chk(self._f["read_ogm"](self._h, _ptr(t), _ptr(r)))
        return {"inst_type": t, "ray_count": r}

    def read_batch_edt(self):
        d = np.empty(self._shape(), np.int32)
        c = np.empty(self._shape() + (3,), np.int32)
        self._chk(self._f["read_batch_edt"](self._h, _ptr(d), _ptr(c)))
        return {"dist_sq": d, "coc": c}

    def read_costmap(self):
        pay = np.empty(self._shape(), SEENDIST_DTYPE)
        hdr = CostMapHdr()
        self._chk(self._f["read_costmap"](self._h, _ptr(pay), C.byref(hdr)))
        return pay, hdr

    def read_costmap_dev(self, dptr):
        """The SeenDist payload into a device buffer (raw device address), asynchronous on the mapper's stream; returns the header."""
        hdr = CostMapHdr()
        self._chk(self._f["read_costmap_dev"](self._h, C.c_void_p(dptr), C.byref(hdr)))
        return hdr

    def costmap_publish(self):
        """Enqueue conversion + asynchronous copy into the library's pinned memory; returns the header at once."""
        hdr = CostMapHdr()
        self._chk(self._f["costmap_publish"](self._h, C.byref(hdr)))
        return hdr

    def costmap_acquire(self, copy=True):
        """Wait for the last publish; the payload as an array over the library's pinned buffer (copy=False: a VIEW that the publish
        after the next one overwrites)."""
        p = C.c_void_p()
        self._chk(self._f["costmap_acquire"](self._h, C.byref(p)))
        buf = (C.c_uint8 * (self.n * SEENDIST_DTYPE.itemsize)).from_address(p.value)
        a = np.frombuffer(buf, dtype=SEENDIST_DTYPE).reshape(self._shape())
        return a.copy() if copy else a

    def query_global_dev(self, d_xyz, n, d_out):
        """n lookups with coordinates (n x 3 int32) and results (n gie_voxel) in DEVICE buffers, enqueued on the mapper's stream."""
        self._chk(self._f["query_global_dev"](self._h, C.c_void_p(d_xyz), int(n), C.c_void_p(d_out)))

    def query_global(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.int32).reshape(-1, 3)
        out = np.empty(xyz.shape[0], VOXEL_DTYPE)
        self._chk(self._f["query_global"](self._h, _ptr(xyz), xyz.shape[0], _ptr(out)))
        return out

    def stats(self):
        s = FrameStats()
        self._chk(self._f["get_stats"](self._h, C.byref(s)))
        return s.as_dict()

    def pivot(self):
        p = (C.c_int32 * 3)()
        self._chk(self._f["get_pivot"](self._h, p))
        return tuple(p)

    # --- tiling (include/gie.h: halo exchange + refinement) ---------------------------------
    def set_tile(self, off, whole):
        o = (C.c_int32 * 3)(*[int(v) for v in off])
        w = (C.c_int32 * 3)(*[int(v) for v in whole])
        self._chk(self._f["set_tile"](self._h, o, w))

    def halo_count(self, face):
        return self._f["halo_count"](self._h, face)

    def halo_export(self, face):
        n = self._f["halo_count"](self._h, face)
        out = np.empty(n, HALO_DTYPE)
        self._chk(self._f["halo_export"](self._h, face, _ptr(out)))
        return out

    def halo_import(self, face, layer):
        layer = np.ascontiguousarray(layer, dtype=HALO_DTYPE)
        assert layer.shape[0] == self._f["halo_count"](self._h, face)
        self._chk(self._f["halo_import"](self._h, face, _ptr(layer)))

    def halo_export_sparse(self, face):
        """The layer's known voxels only: array of HALO_ENTRY_DTYPE (index in the layer, record), in no particular order."""
        n = self._f["halo_count"](self._h, face)
        out = np.empty(n, HALO_ENTRY_DTYPE)
        cnt = C.c_int32(0)
        self._chk(self._f["halo_export_sparse"](self._h, face, _ptr(out), C.byref(cnt)))
        return out[:cnt.value].copy()

    def halo_import_sparse(self, face, entries):
        entries = np.ascontiguousarray(entries, dtype=HALO_ENTRY_DTYPE)
        self._chk(self._f["halo_import_sparse"](self._h, face, _ptr(entries), int(entries.shape[0])))

    def refine(self):
        n = C.c_int32(0)
        self._chk(self._f["refine"](self._h, C.byref(n)))
        return n.value

    def refine_async(self):
        self._chk(self._f["refine"](self._h, None))

    def halo_export_all_dev(self, dptrs):
        """dptrs: {face: device pointer}; all faces in one launch."""
        arr = (C.c_void_p * 6)(*[dptrs.get(f) for f in range(6)])
        self._chk(self._f["halo_export_all_dev"](self._h, arr))

    def halo_import_all_dev(self, dptrs):
        arr = (C.c_void_p * 6)(*[dptrs.get(f) for f in range(6)])
        self._chk(self._f["halo_import_all_dev"](self._h, arr))

    # exchange rounds gated on the device (include/gie.h gie_round_gate ...): pointers are raw device addresses
    def round_gate(self, d_go):
        self._chk(self._f["round_gate"](self._h, C.c_void_p(d_go) if d_go else None))

    def refine_dev(self, d_changed):
        self._chk(self._f["refine_dev"](self._h, C.c_void_p(d_changed)))

    def round_end(self, d_go):
        self._chk(self._f["round_end"](self._h, C.c_void_p(d_go) if d_go else None))

    def round_stats(self):
        """{rounds_enqueued, rounds_run, updates, updates_unconverged} since the mapper was created (synchronises)."""
        v = (C.c_int64 * 4)()
        self._chk(self._f["round_stats"](self._h, v))
        return {"rounds_enqueued": int(v[0]), "rounds_run": int(v[1]), "updates": int(v[2]), "updates_unconverged": int(v[3])}

    # --- changed-block streaming (GlbHashMap::streamPipeline, glb_hash_map.cu:209-247) --------
    def stream_enable(self, on=True):
        self._chk(self._f["stream_enable"](self._h, 1 if on else 0))

    def stream_count(self):
        n = C.c_int32(0)
        self._chk(self._f["stream_changed"](self._h, None, None, 0, C.byref(n)))
        return n.value

    def stream_changed(self, max_blocks=None):
        """(keys [n,3] int32, blocks [n,512] VOXEL_DTYPE in get_voxID_in_VB order, flagged-before-call)."""
        total = self.stream_count()
        n = total if max_blocks is None else min(total, int(max_blocks))
        keys = np.empty((n, 3), np.int32)
        blocks = np.empty((n, 512), VOXEL_DTYPE)
        got = C.c_int32(0)
        if n:
            self._chk(self._f["stream_changed"](self._h, _ptr(keys), _ptr(blocks), n, C.byref(got)))
        return keys, blocks, total

    # --- VOLMAPNODE::publishMap call order (volumetric_mapper.cpp:138-224) -------------
    def update(self, pos, quat_wxyz, sensor_kind, sensor_data, tiled=False, **kw):
        """One map update.  tiled=True stops after the first half of the merge (gie_merge_begin_tiled): the exchange
        functions of gie.tiling export / import the face layers and finish the merge (gie_merge_end)."""
        self.set_pose(pos, quat_wxyz)
        if sensor_kind == "depth":
            self.ogm_depth(sensor_data, **kw)
        elif sensor_kind == "scan2d":
            self.ogm_scan2d(sensor_data, **kw)
        elif sensor_kind == "multiscan":
            self.ogm_multiscan(sensor_data, **kw)
        elif sensor_kind == "pointcloud":
            self.ogm_pointcloud(sensor_data)
        elif sensor_kind == "labels":
            self.ogm_labels(sensor_data)
        else:
            raise ValueError(sensor_kind)
        self.fuse()
        self.batch_edt()
        if tiled:
            self.merge_begin_tiled()
        else:
            self.merge()


_lib = None
_fns = None


def load_library(path=None):
    """Load the HIP C-ABI library. Raises if it has not been built (no fallback)."""
    global _lib, _fns
    if _fns is None:
        p = path or os.environ.get("GIE_LIB") or LIB_PATH    # GIE_LIB: another build of the same HIP library (ablation builds of tools/)
        if not os.path.exists(p):
            raise RuntimeError("%s not found: build it with __graft_entry__.build() "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64,
        # and a process that initialises the system runtime first (through this library) leaves torch
        # without a device ("No HIP GPUs are available").  Loaded the other way round, this library
        # binds to the runtime torch brought.  So: if torch is installed, let it load first.
        if "torch" not in sys.modules and not os.environ.get("GIE_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _lib = C.CDLL(p)
        _fns = _capi.bind(_lib, "gie_", _capi.DEVICE_ONLY)
    return _fns


class Mapper(MapperBase):
    """The MI355X mapper (HIP kernels behind libgie_hip.so)."""

    def __init__(self, cfg):
        super().__init__(load_library(), cfg)

    def profile_enable(self, on=True):
        self._chk(self._f["profile_enable"](self._h, int(on)))

    def profile_read(self):
        """{kernel name: (total_ms, launches)} since the last read (synchronises)."""
        buf = (_capi.KernelTime * 32)()
        n = self._f["profile_read"](self._h, buf, 32)
        if n < 0:
            raise RuntimeError(self._err())
        return {buf[i].name.decode(): (buf[i].total_ms, buf[i].launches) for i in range(n)}

    def stream_handle(self):
        """The mapper's HIP stream as an integer (for torch.cuda.ExternalStream)."""
        p = C.c_void_p()
        self._chk(self._f["get_stream"](self._h, C.byref(p)))
        return p.value or 0

    def halo_export_sparse_dev(self, face, dptr, dcount):
        self._chk(self._f["halo_export_sparse_dev"](self._h, face, C.c_void_p(dptr), C.c_void_p(dcount)))

    def halo_import_sparse_dev(self, face, dptr, dcount):
        self._chk(self._f["halo_import_sparse_dev"](self._h, face, C.c_void_p(dptr), C.c_void_p(dcount)))

    def halo_export_dev(self, face, dptr):
        self._chk(self._f["halo_export_dev"](self._h, face, C.c_void_p(dptr)))

    def halo_import_dev(self, face, dptr):
        self._chk(self._f["halo_import_dev"](self._h, face, C.c_void_p(dptr)))

    # device-resident sensor frames (pointers are raw device addresses, e.g. torch .data_ptr())
    def ogm_depth_dev(self, dptr, rows, cols, cx, cy, fx, fy, valid_nan=False):
        p = CamParam(rows, cols, cx, cy, fx, fy, int(valid_nan))
        self._chk(self._f["ogm_depth_dev"](self._h, C.c_void_p(dptr), C.byref(p)))

    def ogm_labels_dev(self, dptr, borrow=False):
        """A device-resident label plane.  borrow=True: gie_fuse may read it in place (returns whether it will): the plane must then
        stay unchanged until this update's fuse / step has run on the mapper's stream."""
        if not borrow:
            self._chk(self._f["ogm_labels_dev"](self._h, C.c_void_p(dptr)))
            return False
        b = C.c_int(0)
        self._chk(self._f["ogm_labels_dev_borrow"](self._h, C.c_void_p(dptr), C.byref(b)))
        return bool(b.value)

    def ogm_pointcloud_dev(self, dptr, n):
        self._chk(self._f["ogm_pointcloud_dev"](self._h, C.c_void_p(dptr), n))

    def ogm_multiscan_dev(self, dptr, scan_num, ring_num, theta_inc, theta_min, phi_inc, phi_min, max_r=100.0):
        p = MultiScanParam(scan_num, ring_num, max_r, theta_inc, theta_min, phi_inc, phi_min)
        self._chk(self._f["ogm_multiscan_dev"](self._h, C.c_void_p(dptr), C.byref(p)))
