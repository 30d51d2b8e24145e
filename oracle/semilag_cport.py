"""ctypes front end of oracle/semilag_c.c (plain-C float64 port).  TEST INFRASTRUCTURE ONLY.

Same contract as :func:`oracle.semilag.extrapolate` for float32 inputs; used for
full-size parity checks (4096^2) and as the multi-threaded ``cpu_baseline``.
"""

import ctypes
import os

import numpy as np

from . import build as _build
from .semilag import _step_sizes

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.LIB if os.path.exists(_build.LIB) else _build.build()
        lib = ctypes.CDLL(path)
        lib.oracle_semilag_f32.restype = ctypes.c_int
        lib.oracle_semilag_f32.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
            ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int,
        ]
        lib.oracle_num_threads.restype = ctypes.c_int
        _lib = lib
    return _lib


def num_threads():
    return int(_load().oracle_num_threads())


def extrapolate(
    precip,
    velocity,
    timesteps,
    outval=np.nan,
    vel_timestep=1,
    displacement_prev=None,
    n_iter=1,
    return_displacement=False,
    interp_order=1,
    nthreads=0,
):
    lib = _load()
    vel = np.ascontiguousarray(velocity, dtype=np.float32)
    _, m, n = vel.shape
    steps = np.ascontiguousarray(_step_sizes(timesteps, vel_timestep))
    T = steps.size
    if precip is not None:
        pr = np.ascontiguousarray(precip, dtype=np.float32)
        if isinstance(outval, str):
            outval = float(np.nanmin(pr))
        out = np.empty((T, m, n), dtype=np.float32)
        pr_p, out_p = pr.ctypes.data, out.ctypes.data
    else:
        out, pr_p, out_p, outval = None, None, None, np.nan
    dprev = None
    if displacement_prev is not None:
        dprev = np.ascontiguousarray(displacement_prev, dtype=np.float64)
    disp = np.empty((2, m, n), dtype=np.float64)
    rc = lib.oracle_semilag_f32(
        pr_p, vel.ctypes.data, m, n, steps.ctypes.data, T, int(n_iter),
        int(interp_order), float(outval),
        None if dprev is None else dprev.ctypes.data, out_p, disp.ctypes.data,
        int(nthreads),
    )
    if rc != 0:
        raise RuntimeError("oracle_semilag_f32 failed")
    if precip is None:
        return None, disp
    return (out, disp) if return_displacement else out
