"""Measurement aids (HBM counter calibration copies, gather cost probe, DPP lane-shift probe).

Built into their own shared object (``python -m tools.calib.build``); nothing here is part of the
product library ``libpysteps_hip.so``.
"""

import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpsh_calib.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            from . import build

            build.build()
        _lib = ctypes.CDLL(LIB_PATH)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        _lib.calib_dpp.argtypes = [vp]
        _lib.calib_gather.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_float)]
        _lib.calib_copy.argtypes = [vp, vp, ctypes.c_size_t, ci]
        _lib.calib_valu.argtypes = [vp, ci, ci, ci, ctypes.POINTER(ctypes.c_float)]
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (code %d)" % (what, rc))
