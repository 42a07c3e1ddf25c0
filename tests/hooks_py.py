"""Loader for the TEST build of the HIP library (tests/gpu_helpers/libgie_hip_test.so = the product's sources compiled with
-DGIE_TEST_HOOKS): the only build whose behaviour can be steered through GIE_* environment switches (forced tile lists, pool
base, staging chunk size, un-completed batch-EDT export ...) and that exports the gie_debug_* hooks.  The product library
(gie.Mapper, libgie_hip.so) has neither (tests/test_capi_symbols.py).  Switches are read once per process and switch: set them
(monkeypatch.setenv) BEFORE the first mapper that needs them is created."""
import ctypes as C
import os

from gie import _capi
from gie.mapper import Mapper, load_library

HERE = os.path.dirname(os.path.abspath(__file__))
TEST_SO = os.path.join(HERE, "gpu_helpers", "libgie_hip_test.so")
_fns = None
_lib = None


def load():
    global _fns, _lib
    if _fns is None:
        if not os.path.exists(TEST_SO):
            import sys
            sys.path.insert(0, os.path.dirname(HERE))
            import __graft_entry__
            __graft_entry__.build_hip_test_hooks()
        load_library()                                  # (torch's HIP runtime first, as the package does it)
        _lib = C.CDLL(TEST_SO)
        _fns = _capi.bind(_lib, "gie_", _capi.DEVICE_ONLY)
        _lib.gie_debug_fault_barrier.argtypes = [C.c_void_p, C.c_int]
        _lib.gie_debug_place_probe.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        _lib.gie_debug_nbr_check.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    return _fns


class HooksMapper(Mapper):
    """gie.Mapper on the test build of the library."""

    def __init__(self, cfg):
        super(Mapper, self).__init__(load(), cfg)

    def debug_fault_barrier(self, updates):
        return _lib.gie_debug_fault_barrier(self._h, int(updates))

    def debug_place_probe(self, reps):
        ms = C.c_float(0)
        rc = _lib.gie_debug_place_probe(self._h, int(reps), C.byref(ms))
        if rc:
            raise RuntimeError(self._err())
        return ms.value

    def debug_nbr_check(self):
        """rows of the neighbour table of waves A / B that disagree with the hash, over every live block"""
        bad = C.c_int32(-1)
        if _lib.gie_debug_nbr_check(self._h, C.byref(bad)):
            raise RuntimeError(self._err())
        return bad.value
