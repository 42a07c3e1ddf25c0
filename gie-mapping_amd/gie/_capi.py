"""ctypes view of include/gie.h.

`bind(lib, prefix)` attaches argtypes/restype for every entry point of the C-ABI to a loaded
shared library whose symbols start with `prefix`.  The product library uses prefix "gie_".
(The test-only CPU oracle exports the same signatures under another prefix and is bound by
tests/, never from this package.)
"""
import ctypes as C

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i8p = C.POINTER(C.c_int8)
c_u8p = C.POINTER(C.c_uint8)


class Config(C.Structure):
    """gie_config (include/gie.h)."""
    _fields_ = [
        ("voxel_width", C.c_float),
        ("local_size", C.c_int32 * 3),
        ("occupancy_threshold", C.c_int32),
        ("ogm_min_h", C.c_float),
        ("ogm_max_h", C.c_float),
        ("cutoff_grids_sq", C.c_int32),
        ("fast_mode", C.c_int32),
        ("for_motion_planner", C.c_int32),
        ("robot_r2_grids", C.c_int32),
        ("max_blocks", C.c_int32),
        ("device_id", C.c_int32),
        ("retain_radius_blocks", C.c_int32),
        ("wave_workgroups", C.c_int32),
        ("place_tries", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class MultiScanParam(C.Structure):
    _fields_ = [("scan_num", C.c_int32), ("ring_num", C.c_int32), ("max_r", C.c_float),
                ("theta_inc", C.c_float), ("theta_min", C.c_float), ("phi_inc", C.c_float),
                ("phi_min", C.c_float)]


class CamParam(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("cx", C.c_float), ("cy", C.c_float),
                ("fx", C.c_float), ("fy", C.c_float), ("valid_nan", C.c_int32)]


class ScanParam(C.Structure):
    _fields_ = [("scan_num", C.c_int32), ("max_r", C.c_float), ("theta_inc", C.c_float),
                ("theta_min", C.c_float)]


class Voxel(C.Structure):
    _fields_ = [("occ_val", C.c_uint8), ("vox_type", C.c_int8), ("pad", C.c_int16),
                ("dist_sq", C.c_int32), ("coc", C.c_int32 * 3)]


class CostMapHdr(C.Structure):
    _fields_ = [("x_size", C.c_int32), ("y_size", C.c_int32), ("z_size", C.c_int32),
                ("x_origin", C.c_float), ("y_origin", C.c_float), ("z_origin", C.c_float),
                ("width", C.c_float), ("type", C.c_uint8), ("pad", C.c_uint8 * 3)]


class FrameStats(C.Structure):
    _fields_ = [("frame", C.c_int32), ("blocks_total", C.c_int32), ("blocks_new", C.c_int32),
                ("seeds_a", C.c_int32), ("seeds_b", C.c_int32), ("seeds_c", C.c_int32),
                ("front_b", C.c_int32), ("front_c", C.c_int32),
                ("visits_a", C.c_int32), ("visits_b", C.c_int32), ("visits_c", C.c_int32),
                ("levels_a", C.c_int32), ("levels_b", C.c_int32), ("levels_c", C.c_int32),
                ("us_ogm", C.c_float), ("us_fuse", C.c_float), ("us_edt", C.c_float),
                ("us_merge", C.c_float),
                ("total_visits_a", C.c_int64), ("total_visits_b", C.c_int64), ("total_visits_c", C.c_int64),
                ("known_tiles", C.c_int32), ("frontier_tiles", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# name -> (restype, argtypes); handle is a void*
_H = C.c_void_p
SIGNATURES = {
    "create": (_H, [C.POINTER(Config)]),
    "destroy": (None, [_H]),
    "set_pose": (C.c_int, [_H, c_f32p, c_f32p]),
    "ogm_pointcloud": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "ogm_multiscan": (C.c_int, [_H, C.c_void_p, C.POINTER(MultiScanParam)]),
    "ogm_depth": (C.c_int, [_H, C.c_void_p, C.POINTER(CamParam)]),
    "ogm_scan2d": (C.c_int, [_H, C.c_void_p, C.POINTER(ScanParam)]),
    "ogm_labels": (C.c_int, [_H, C.c_void_p]),
    "set_ext_boxes": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "fuse": (C.c_int, [_H]),
    "batch_edt": (C.c_int, [_H]),
    "merge": (C.c_int, [_H]),
    "merge_begin_tiled": (C.c_int, [_H]),
    "merge_end": (C.c_int, [_H]),
    "step": (C.c_int, [_H]),
    "read_local": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "read_ogm": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "read_batch_edt": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "read_costmap": (C.c_int, [_H, C.c_void_p, C.POINTER(CostMapHdr)]),
    "query_global": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p]),
    "get_stats": (C.c_int, [_H, C.POINTER(FrameStats)]),
    "get_pivot": (C.c_int, [_H, c_i32p]),
    "set_tile": (C.c_int, [_H, c_i32p, c_i32p]),
    "halo_count": (C.c_int, [_H, C.c_int]),
    "halo_export": (C.c_int, [_H, C.c_int, C.c_void_p]),
    "halo_import": (C.c_int, [_H, C.c_int, C.c_void_p]),
    "refine": (C.c_int, [_H, c_i32p]),
    "stream_enable": (C.c_int, [_H, C.c_int]),
    "stream_changed": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, c_i32p]),
}
class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("total_ms", C.c_float), ("launches", C.c_int32)]


# exchange rounds gated on the device: the device library and the test-only emulation of its logic (not the oracle, which has no device)
ROUND_API = {
    "read_costmap_dev": (C.c_int, [_H, C.c_void_p, C.POINTER(CostMapHdr)]),
    "costmap_publish": (C.c_int, [_H, C.POINTER(CostMapHdr)]),
    "costmap_acquire": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "query_global_dev": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p]),
    "round_gate": (C.c_int, [_H, C.c_void_p]),
    "refine_dev": (C.c_int, [_H, C.c_void_p]),
    "round_end": (C.c_int, [_H, C.c_void_p]),
    "round_stats": (C.c_int, [_H, C.POINTER(C.c_int64)]),
    "halo_export_all_dev": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "halo_import_all_dev": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
}

# entry points only the device library has
DEVICE_ONLY = {
    "profile_enable": (C.c_int, [_H, C.c_int]),
    "profile_read": (C.c_int, [_H, C.POINTER(KernelTime), C.c_int]),
    "last_error": (C.c_char_p, []),
    "sync": (C.c_int, [_H]),
    "get_stream": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "ogm_pointcloud_dev": (C.c_int, [_H, C.c_void_p, C.c_int]),
    "halo_export_dev": (C.c_int, [_H, C.c_int, C.c_void_p]),
    "halo_import_dev": (C.c_int, [_H, C.c_int, C.c_void_p]),
    "halo_export_sparse": (C.c_int, [_H, C.c_int, C.c_void_p, c_i32p]),
    "halo_import_sparse": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int32]),
    "halo_export_sparse_dev": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "halo_import_sparse_dev": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "ogm_multiscan_dev": (C.c_int, [_H, C.c_void_p, C.POINTER(MultiScanParam)]),
    "ogm_depth_dev": (C.c_int, [_H, C.c_void_p, C.POINTER(CamParam)]),
    "ogm_labels_dev": (C.c_int, [_H, C.c_void_p]),
    "ogm_labels_dev_borrow": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_int)]),
}
DEVICE_ONLY.update(ROUND_API)


def bind(lib, prefix, extra=None):
    """Return {short_name: ctypes function} for every symbol `prefix+name`."""
    out = {}
    table = dict(SIGNATURES)
    if extra:
        table.update(extra)
    for name, (res, args) in table.items():
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = args
        out[name] = fn
    return out
