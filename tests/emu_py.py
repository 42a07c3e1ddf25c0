"""Loader for the test-only sequential emulation of the device logic (tests/emu)."""
import ctypes as C
import os
import subprocess

from gie import _capi
from gie.mapper import MapperBase

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
EMU_SO = os.path.join(EMU_DIR, "libgie_emu.so")
CSRC = os.path.join(os.path.dirname(HERE), "gie-mapping_amd", "csrc")
_fns = None
_lib = None


def load():
    global _fns
    if _fns is None:
        srcs = [os.path.join(EMU_DIR, "gie_emu.cpp"), os.path.join(EMU_DIR, "gie_emu_ops.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
        newest = max(os.path.getmtime(s) for s in srcs)
        import fcntl
        with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:       # pytest-xdist workers: one builds, the others wait
            fcntl.flock(lock, fcntl.LOCK_EX)
            if (not os.path.exists(EMU_SO)) or os.path.getmtime(EMU_SO) < newest:
                tmp = EMU_SO + ".%d.tmp" % os.getpid()
                subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", tmp,
                                       os.path.join(EMU_DIR, "gie_emu.cpp")])
                os.replace(tmp, EMU_SO)
        global _lib
        lib = _lib = C.CDLL(EMU_SO)
        _fns = _capi.bind(lib, "gie_", {**_capi.ROUND_API, "last_error": (C.c_char_p, []), "sync": (C.c_int, [C.c_void_p]),
                                        "halo_export_sparse": _capi.DEVICE_ONLY["halo_export_sparse"],
                                        "halo_import_sparse": _capi.DEVICE_ONLY["halo_import_sparse"]})
    return _fns


def wave_c_model(device, r0filter=False):
    """Which statement of wave C the emulation runs from now on: the canonical one (device=False) or the sequential model of the
    DEVICE's tile rounds (gie_emu.cpp be_wave_c_device); r0filter=True puts the round-0 halo filter back that round 4 removed."""
    load()
    _lib.gie_emu_wave_c_model(int(device), int(bool(r0filter)))       # device = 2: halos read the live plane (the other end of the device's race)


class EmuMapper(MapperBase):
    def __init__(self, cfg):
        super().__init__(load(), cfg)

    def debug_nbr_check(self):
        """rows of the neighbour table of waves A / B that disagree with the hash, over every live block (test hook)"""
        bad = C.c_int32(-1)
        _lib.gie_debug_nbr_check.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        if _lib.gie_debug_nbr_check(self._h, C.byref(bad)):
            raise RuntimeError(self._err())
        return bad.value
