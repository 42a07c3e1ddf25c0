"""ctypes loader for the CPU oracle (oracle/libgie_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from gie import _capi
from gie.mapper import MapperBase

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libgie_oracle.so")

_fns = None
_lib = None


def load():
    global _fns, _lib
    if _fns is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("gie_oracle.c", "edt_mt.c")]
        import fcntl
        with open(os.path.join(ORACLE_DIR, ".build.lock"), "w") as lock:    # pytest-xdist workers: one builds, the others wait
            fcntl.flock(lock, fcntl.LOCK_EX)
            if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(s) for s in srcs):
                subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
        _lib = C.CDLL(ORACLE_SO)
        _fns = _capi.bind(_lib, "go_")
        _lib.go_brute_force_edt.restype = C.c_int
        _lib.go_brute_force_edt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        _lib.go_edt_mt.restype = C.c_int
        _lib.go_edt_mt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.go_edt_only.restype = C.c_int
        _lib.go_edt_only.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _fns


class OracleMapper(MapperBase):
    def __init__(self, cfg):
        super().__init__(load(), cfg)

    def edt_only(self, glb_type):
        t = np.ascontiguousarray(glb_type, dtype=np.int8)
        d = np.empty(self._shape(), np.int32)
        c = np.empty(self._shape() + (3,), np.int32)
        _lib.go_edt_only(self._h, t.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p),
                         c.ctypes.data_as(C.c_void_p))
        return d, c


def brute_force_edt(occ):
    load()
    occ = np.ascontiguousarray(occ, dtype=np.int8)
    Z, Y, X = occ.shape
    out = np.empty((Z, Y, X), np.int32)
    _lib.go_brute_force_edt(occ.ctypes.data_as(C.c_void_p), X, Y, Z, out.ctypes.data_as(C.c_void_p))
    return out


EDT_MT_NONE = 0x3fffffff


def edt_mt(types, nthreads=0, want_coc=True):
    """Multi-threaded exact separable EDT (oracle/edt_mt.c): the full-size CPU baseline.
    types: int8 [Z][Y][X], 2 = occupied.  Returns (dist_sq, coc_packed or None)."""
    load()
    t = np.ascontiguousarray(types, dtype=np.int8)
    Z, Y, X = t.shape
    d = np.empty((Z, Y, X), np.int32)
    c = np.empty((Z, Y, X), np.int32) if want_coc else None
    rc = _lib.go_edt_mt(t.ctypes.data_as(C.c_void_p), X, Y, Z, int(nthreads) if nthreads else (os.cpu_count() or 1),
                        d.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p) if want_coc else None)
    if rc != 0:
        raise RuntimeError("go_edt_mt failed (%d)" % rc)
    return d, c
