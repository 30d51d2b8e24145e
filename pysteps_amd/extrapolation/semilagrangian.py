"""Semi-Lagrangian backward extrapolation on MI355X (HIP), drop-in for
``pysteps.extrapolation.semilagrangian.extrapolate``
(reference: pysteps/extrapolation/semilagrangian.py:21-266).

Same signature, argument meaning, return values and exceptions as the reference;
the trajectory integration and the resampling run in one fused HIP kernel
(``csrc/semilag.hip``) reached through the C ABI ``psh_semilag_*``
(``include/pysteps_hip.h``).  There is no CPU fallback.

Differences, all documented in DESIGN.md:

* arithmetic is float32 on device (the reference computes in float64 inside
  SciPy); the advected field is within 1e-4 relative L2 of the reference (measured
  ~1e-6) and the returned displacement is float64 like the reference's;
* a custom ``xy_coords`` grid (:174-179) rides into the kernel as base-position offsets in the displacement buffer
  (``_grid_offsets``); every ``interp_order`` scipy.ndimage accepts (0 .. 5) and every boundary mode is native;
* ``precip``/``velocity``/``displacement_prev`` may also be
  :class:`pysteps_amd.device.DeviceArray` objects; then nothing crosses PCIe and
  the results are DeviceArrays too (used by the resident nowcast loop and bench).
"""

import ctypes
import time
import warnings
import weakref

import numpy as np

from .. import _lib, _pinned
from ..device import DeviceArray

__all__ = ["extrapolate"]


# psh_semilag_host flags / input status bits (include/pysteps_hip.h PSH_SL_*)
_FLAG_ALLOW_NONFINITE, _FLAG_OUTVAL_MIN, _FLAG_PRECIP_F64, _FLAG_VELOCITY_F64, _FLAG_OUT_F64 = 1, 2, 4, 8, 16
_FLAG_BASE_IN_DISP = 32
_ST_PRECIP_NONFINITE, _ST_PRECIP_ALL_NONFINITE, _ST_VELOCITY_NONFINITE, _ST_VELOCITY_ALL_NONFINITE = 1, 2, 4, 8

# scipy.ndimage boundary modes of the field resampling -> PSH_MODE_* (include/pysteps_hip.h)
_BOUNDARY_MODES = {"constant": 0, "nearest": 1, "reflect": 2, "mirror": 3, "wrap": 4,
                   "grid-constant": 5, "grid-wrap": 6}


def _reference_extrapolate():
    try:
        from pysteps.extrapolation.semilagrangian import extrapolate as ref  # noqa: PLC0415
    except Exception:
        return None
    return None if ref is extrapolate else ref


def _unsupported(what, args, kwargs):
    ref = _reference_extrapolate()
    if ref is None:
        raise NotImplementedError(
            "pysteps_amd semilagrangian: %s is not implemented on the HIP path and the "
            "reference pysteps implementation is not importable" % what
        )
    warnings.warn("pysteps_amd semilagrangian: %s -> delegating to the reference CPU path" % what)
    return ref(*args, **kwargs)


_verified_grids = []  # [(weakref to the array, data pointer, shape)]: grids already compared in full


def _grid_sample_ok(xy, m, n):
    cols, rows = np.arange(n), np.arange(m)
    for r in (0, m - 1):
        if not (np.array_equal(xy[0, r], cols) and np.all(xy[1, r] == r)):
            return False
    for c in (0, n - 1):
        if not (np.array_equal(xy[1, :, c], rows) and np.all(xy[0, :, c] == c)):
            return False
    sr, sc = max(1, m // 61), max(1, n // 67)
    return bool(np.array_equal(xy[0, ::sr, ::sc], np.broadcast_to(cols[::sc], (len(rows[::sr]), len(cols[::sc]))))
                and np.array_equal(xy[1, ::sr, ::sc], np.broadcast_to(rows[::sr, None], (len(rows[::sr]), len(cols[::sc])))))


def _is_default_grid(xy_coords, m, n):
    """True if xy_coords is the integer meshgrid the reference builds itself (:174-179).

    The whole array is compared (every element, not a sample).  The callers that pass
    ``xy_coords`` (nowcasts/utils.py:361-365, steps.py:661-662) build it once and hand the SAME
    array to every call, so an array that has been verified in full is remembered by identity (weak
    reference + data pointer + shape) and afterwards only sampled (``_grid_sample_ok``) - the full
    scan of a (2, 4096, 4096) int64 grid costs more than the advection itself."""
    xy = np.asarray(xy_coords)
    if xy.shape != (2, m, n):
        return False
    key = (xy.__array_interface__["data"][0], xy.shape, xy.dtype.str)
    for ref, k in _verified_grids:
        if ref() is xy_coords and k == key:
            # remembered by identity - but the caller may have written into the array since: the
            # border rows / columns and a strided interior sample are compared again on every call
            # (O(m + n) + 4096 elements instead of 2 m n)
            return _grid_sample_ok(xy, m, n)
    ok = bool(np.array_equal(xy[0], np.broadcast_to(np.arange(n), (m, n)))
              and np.array_equal(xy[1], np.broadcast_to(np.arange(m)[:, None], (m, n))))
    if ok:
        try:
            _verified_grids.append((weakref.ref(xy_coords), key))
            del _verified_grids[:-8]
        except TypeError:
            pass  # not weak-referenceable (a list): verified every time
    return ok


def _grid_offsets(xy_coords, m, n):
    """``xy_coords - meshgrid`` as float64 (2, m, n): where the trajectories of a custom grid start, relative to the
    integer grid the kernels index (reference :174-179, :182, :221: every coordinate is ``xy_coords + displacement``).
    The kernels split it into integer position + float32 fraction (6e-8 px), as they split a displacement_prev."""
    xy = np.asarray(xy_coords, dtype=np.float64)
    if xy.shape != (2, m, n):
        raise ValueError("xy_coords must have shape (2, %d, %d)" % (m, n))
    off = np.empty((2, m, n), dtype=np.float64)
    np.subtract(xy[0], np.arange(n, dtype=np.float64)[None, :], out=off[0])
    np.subtract(xy[1], np.arange(m, dtype=np.float64)[:, None], out=off[1])
    return off


def _step_increments(timesteps, vel_timestep):
    """timestep_diff / vel_timestep (reference :159-165, :198)."""
    if isinstance(timesteps, int) and not isinstance(timesteps, bool):
        if timesteps < 1:
            raise ValueError("timesteps must be a positive integer")
        return np.ones(timesteps, dtype=np.float64)  # vel_timestep forced to 1 (:161)
    ts = np.asarray(timesteps, dtype=np.float64).ravel()
    if ts.size == 0:
        raise ValueError("timesteps is empty")
    if np.any(np.diff(ts) <= 0.0):
        raise ValueError("the given timestep sequence is not monotonously increasing")
    return np.concatenate([ts[:1], np.diff(ts)]) / float(vel_timestep)


def extrapolate(
    precip,
    velocity,
    timesteps,
    outval=np.nan,
    xy_coords=None,
    allow_nonfinite_values=False,
    vel_timestep=1,
    **kwargs,
):
    """Apply semi-Lagrangian backward extrapolation to a 2-d precipitation field.

    Parameters, other parameters (``displacement_prev``, ``n_iter``,
    ``return_displacement``, ``vel_timestep``, ``interp_order``,
    ``map_coordinates_mode``, ``verbose``) and returns are those of the
    reference (semilagrangian.py:31-103): ``(num_timesteps, m, n)`` array, or
    ``(array, displacement)`` / ``(None, displacement)`` with
    ``return_displacement=True``.
    """
    call_args = (precip, velocity, timesteps)
    call_kwargs = dict(
        outval=outval, xy_coords=xy_coords, allow_nonfinite_values=allow_nonfinite_values,
        vel_timestep=vel_timestep, **kwargs,
    )
    on_device = isinstance(velocity, DeviceArray)
    if (precip is not None and isinstance(precip, DeviceArray) != on_device) or (
            kwargs.get("displacement_prev") is not None
            and isinstance(kwargs["displacement_prev"], DeviceArray) != on_device):
        raise ValueError("precip, velocity and displacement_prev must all be NumPy arrays or all be DeviceArrays")

    if precip is not None and precip.ndim != 2:
        raise ValueError("precip must be a two-dimensional array")
    if velocity.ndim != 3:
        raise ValueError("velocity must be a three-dimensional array")

    # the non-finite checks of the reference (:106-137) run as device reductions inside
    # psh_semilag_host, after the upload (a NumPy isfinite scan of three 4096^2 planes costs more
    # than the whole kernel); same exceptions, raised below
    if isinstance(timesteps, list) and not sorted(timesteps) == timesteps:
        raise ValueError("timesteps is not in ascending order")

    verbose = kwargs.get("verbose", False)
    displacement_prev = kwargs.get("displacement_prev", None)
    n_iter = int(kwargs.get("n_iter", 1))
    return_displacement = kwargs.get("return_displacement", False)
    interp_order = kwargs.get("interp_order", 1)
    map_coordinates_mode = kwargs.get("map_coordinates_mode", "constant")

    if precip is None and not return_displacement:
        raise ValueError("precip is None but return_displacement is False")
    if "D_prev" in kwargs:
        warnings.warn("deprecated argument D_prev is ignored, use displacement_prev instead")

    steps = _step_increments(timesteps, vel_timestep)

    if velocity.shape[0] != 2:
        raise ValueError("velocity must have shape (2, m, n)")
    m, n = velocity.shape[1:]
    if precip is not None and tuple(precip.shape) != (m, n):
        raise ValueError("precip and velocity have incompatible shapes")

    # ---- options outside the kernel's contract -------------------------
    if interp_order not in (0, 1, 2, 3, 4, 5):  # (what scipy.ndimage accepts; anything else is the reference's error to raise)
        return _unsupported("interp_order=%r" % (interp_order,), call_args, call_kwargs)
    if map_coordinates_mode not in _BOUNDARY_MODES:
        raise RuntimeError("boundary mode not supported")  # what scipy.ndimage raises
    # the boundary mode rides in the second byte of the interp_order word (include/pysteps_hip.h)
    interp_order = int(interp_order) | (_BOUNDARY_MODES[map_coordinates_mode] << 8)
    # a custom grid: its base positions relative to the integer grid go where a displacement_prev would (the kernels
    # start every trajectory at integer position + fraction anyway); the displacement that comes back is made
    # relative to xy_coords again
    grid_off = None
    if xy_coords is not None and not _is_default_grid(xy_coords, m, n):
        grid_off = _grid_offsets(xy_coords, m, n)
    if n_iter < 0:
        n_iter = 0  # the reference treats any n_iter <= 0 as "no midpoint rule" (:211-219)

    if verbose:
        print("Computing the advection with the semi-lagrangian scheme.")
        t0 = time.time()

    lib = _lib.lib()
    T = int(steps.size)

    if on_device:
        result = _run_device(lib, precip, velocity, steps, outval, displacement_prev, n_iter,
                             return_displacement, interp_order, grid_off)
    else:
        flags = _FLAG_ALLOW_NONFINITE if allow_nonfinite_values else 0
        if isinstance(outval, str):
            if outval != "min":
                raise ValueError("outval must be a number or 'min'")
            flags |= _FLAG_OUTVAL_MIN  # np.nanmin(precip), as a device reduction (:171-172)
            outval = float("nan")
        # float32 and float64 arrays go to the device as they are (float64 is narrowed there);
        # the advected field comes back in the dtype of precip, like SciPy's output (:225-232)
        def _as_input(a):
            a = np.asarray(a)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64 if a.dtype.itemsize > 4 else np.float32)
            return np.ascontiguousarray(a)

        vin = _as_input(velocity)
        pin = None if precip is None else _as_input(np.ma.getdata(precip) if np.ma.isMaskedArray(precip) else precip)
        if vin.dtype == np.float64:
            flags |= _FLAG_VELOCITY_F64
        out_dtype = None
        if pin is not None:
            out_dtype = np.asarray(precip).dtype
            if pin.dtype == np.float64:
                flags |= _FLAG_PRECIP_F64
            if out_dtype == np.float64:
                flags |= _FLAG_OUT_F64
        dprev = None
        if displacement_prev is not None:
            dprev = np.ascontiguousarray(displacement_prev, dtype=np.float64)
            if dprev.shape != (2, m, n):
                raise ValueError("displacement_prev must have shape (2, m, n)")
        if grid_off is not None:
            if dprev is None:
                dprev = grid_off
                flags |= _FLAG_BASE_IN_DISP  # no previous displacement: the first increment is the grid's velocity (:203)
            else:
                dprev = dprev + grid_off
        # results on pinned blocks of the library's pool: the device-to-host copies land in the
        # arrays the caller receives (csrc/hostpath.hip)
        out = None
        if pin is not None:
            out = _pinned.empty((T, m, n), np.float64 if flags & _FLAG_OUT_F64 else np.float32)
        disp = _pinned.empty((2, m, n), np.float64) if return_displacement else None
        status = ctypes.c_int(0)
        rc = lib.psh_semilag_host(
            None if pin is None else pin.ctypes.data, vin.ctypes.data, m, n,
            steps.ctypes.data, T, n_iter, int(interp_order),
            float(outval) if precip is not None else float("nan"),
            None if dprev is None else dprev.ctypes.data,
            None if disp is None else disp.ctypes.data,
            None if out is None else out.ctypes.data, flags, ctypes.byref(status),
        )
        if rc == _lib.PSH_EINPUT:  # the reference's messages, in the reference's order (:106-125)
            st = status.value
            if not allow_nonfinite_values and st & _ST_PRECIP_NONFINITE:
                raise ValueError("precip contains non-finite values")
            if not allow_nonfinite_values and st & _ST_VELOCITY_NONFINITE:
                raise ValueError("velocity contains non-finite values")
            if st & _ST_PRECIP_ALL_NONFINITE:
                raise ValueError("precip contains only non-finite values")
            raise ValueError("velocity contains only non-finite values")
        _lib.check(rc, "psh_semilag_host")
        if grid_off is not None and disp is not None:
            np.subtract(disp, grid_off, out=disp)
        if out is not None and out.dtype != out_dtype and np.issubdtype(out_dtype, np.floating):
            out = out.astype(out_dtype)  # float16 / longdouble inputs
        if precip is None:
            result = (None, disp)
        elif return_displacement:
            result = (out, disp)
        else:
            result = out

    if verbose:
        print("--- %s seconds ---" % (time.time() - t0))
    return result


def _run_device(lib, precip, velocity, steps, outval, displacement_prev, n_iter,
                return_displacement, interp_order, grid_off=None):
    """All operands resident in HBM; asynchronous on the library stream."""
    m, n = velocity.shape[1:]
    T = int(steps.size)
    if velocity.dtype != np.float32 or (precip is not None and precip.dtype != np.float32):
        raise ValueError("device-resident precip/velocity must be float32")
    if precip is not None and not isinstance(precip, DeviceArray):
        raise ValueError("precip must be a DeviceArray when velocity is one")
    if isinstance(outval, str):
        if outval != "min":
            raise ValueError("outval must be a number or 'min'")
        from ..utils.transformation import field_stats  # noqa: PLC0415

        outval = field_stats(precip)[0] if precip is not None else float("nan")  # device reduction
    disp = None
    resume = 0
    if displacement_prev is not None:
        if not isinstance(displacement_prev, DeviceArray) or displacement_prev.dtype != np.float64:
            raise ValueError("displacement_prev must be a float64 DeviceArray on the device path")
        if displacement_prev.shape != (2, m, n):
            raise ValueError("displacement_prev must have shape (2, m, n)")
        disp = DeviceArray((2, m, n), np.float64)  # inputs are never mutated (:205)
        _lib.check(lib.psh_memcpy_d2d(disp.ptr, displacement_prev.ptr, disp.nbytes), "d2d")
        resume = 1
    elif return_displacement or grid_off is not None:
        disp = DeviceArray((2, m, n), np.float64)
    off_d = None
    if grid_off is not None:
        # base positions of a custom xy_coords grid: into the displacement buffer (added to a displacement_prev)
        off_d = DeviceArray.from_host(grid_off)
        if resume:
            _lib.check(lib.psh_axpy_f64_dev(disp.ptr, off_d.ptr, 1.0, disp.size), "psh_axpy_f64_dev")
        else:
            _lib.check(lib.psh_memcpy_d2d(disp.ptr, off_d.ptr, disp.nbytes), "d2d")
            resume = 2  # PSH_SL_RESUME_BASE: positions from the buffer, the first increment from the grid's velocity
    out = None if precip is None else DeviceArray((T, m, n), np.float32)
    # a motion field that comes with its {u, v}-interleaved twin (dense_lucaskanade on resident frames) is
    # gathered from that twin directly
    pairs = getattr(velocity, "uv_pairs", None)
    if pairs is not None and (pairs.shape != (m, n, 2) or pairs.dtype != np.float32):
        pairs = None
    rc = lib.psh_semilag_uv_dev(
        None if precip is None else precip.ptr, velocity.ptr, None if pairs is None else pairs.ptr, m, n,
        steps.ctypes.data, T, n_iter, int(interp_order), float(outval) if precip is not None else float("nan"),
        None if disp is None else disp.ptr, resume, None if out is None else out.ptr,
    )
    _lib.check(rc, "psh_semilag_uv_dev")
    if off_d is not None:
        if return_displacement or precip is None:
            _lib.check(lib.psh_axpy_f64_dev(disp.ptr, off_d.ptr, -1.0, disp.size), "psh_axpy_f64_dev")
        else:
            disp = None
    if precip is None:
        return None, disp
    return (out, disp) if return_displacement else out
