"""Sparse -> dense interpolation on MI355X, mirror of pysteps/utils/interpolate.py.

``idwinterp2d`` has the reference's signature and return convention
(interpolate.py:26-114) including the input checks and trivial cases its
``prepare_interpolator`` decorator performs (decorators.py:153-250); the k-NN search
and the weighting run in the HIP kernel ``csrc/idw.hip`` through ``psh_idw_*``.
"""

import ctypes
import warnings

import numpy as np

from .. import _lib
from ..device import DeviceArray

__all__ = ["idwinterp2d"]


def _reference_idw():
    try:
        from pysteps.utils.interpolate import idwinterp2d as ref  # noqa: PLC0415
    except Exception:
        return None
    return None if ref is idwinterp2d else ref


def _regular_axis(grid, name):
    """(origin, spacing) of a regularly spaced 1-d axis."""
    grid = np.asarray(grid, dtype=float)
    if grid.ndim != 1 or grid.size == 0:
        raise ValueError("%s must be a non-empty 1-d array" % name)
    if grid.size == 1:
        return float(grid[0]), 1.0
    steps = np.diff(grid)
    if steps[0] == 0 or not np.allclose(steps, steps[0], rtol=1e-9, atol=0.0):
        return None
    return float(grid[0]), float(steps[0])


def _check_inputs(xy_coord, values):
    """Validation of decorators.py:165-198 (same messages)."""
    values = np.array(values, dtype=float, copy=True)
    xy_coord = np.array(xy_coord, dtype=float, copy=True)
    if np.any(~np.isfinite(values)):
        raise ValueError("argument 'values' contains non-finite values")
    if np.any(~np.isfinite(xy_coord)):
        raise ValueError("argument 'xy_coord' contains non-finite values")
    if values.ndim > 2:
        raise ValueError(
            "argument 'values' must have 1 (n) or 2 dimensions (n, m), "
            f"but it has {values.ndim}"
        )
    if not xy_coord.ndim == 2:
        raise ValueError(
            "argument 'xy_coord' must have 2 dimensions (n, 2), " f"but it has {xy_coord.ndim}"
        )
    if not values.shape[0] == xy_coord.shape[0]:
        raise ValueError(
            "the number of samples in argument 'values' does not match the "
            f"number of coordinates {values.shape[0]}!={xy_coord.shape[0]}"
        )
    return xy_coord, values


def idw_to_device(xy_coord, values2, m, n, x0=0.0, dx=1.0, y0=0.0, dy=1.0, power=0.5, k=20,
                  dist_offset=0.5):
    """(L,2) samples -> float32 DeviceArray (2,m,n); asynchronous on the library stream."""
    lib = _lib.lib()
    xy32 = np.ascontiguousarray(xy_coord, dtype=np.float32)
    uv32 = np.ascontiguousarray(values2, dtype=np.float32)
    L = xy32.shape[0]
    k_eff = L if k is None else int(min(k, L))
    xs = (x0, x0 + dx * (n - 1), float(xy_coord[:, 0].min()), float(xy_coord[:, 0].max()))
    ys = (y0, y0 + dy * (m - 1), float(xy_coord[:, 1].min()), float(xy_coord[:, 1].max()))
    reach = float(np.hypot(max(xs) - min(xs), max(ys) - min(ys))) * 1.001 + 1.0
    d_xy, d_uv = DeviceArray.from_host(xy32, sync=False), DeviceArray.from_host(uv32, sync=False)
    out = DeviceArray((2, m, n), np.float32)
    _lib.check(
        lib.psh_idw_dev(d_xy.ptr, d_uv.ptr, L, m, n, x0, dx, y0, dy, k_eff, float(power),
                        float(dist_offset), reach, out.ptr),
        "psh_idw_dev",
    )
    out._keep = (d_xy, d_uv)  # the sample buffers must outlive the queued kernel
    return out


def idwinterp2d(xy_coord, values, xgrid, ygrid, power=0.5, k=20, dist_offset=0.5, **kwargs):
    """Inverse distance weighting interpolation of a sparse (multivariate) array.

    Parameters and return value as in the reference (interpolate.py:31-66):
    ``(ygrid.size, xgrid.size)`` for 1-d ``values`` or ``(m, ygrid.size, xgrid.size)``
    float64.  ``nchunks``/``hkey`` keyword arguments are accepted and ignored (the
    result does not depend on the chunking the reference uses to bound memory).
    """
    xy_coord, values = _check_inputs(xy_coord, values)
    xgrid, ygrid = np.asarray(xgrid), np.asarray(ygrid)
    grid_shape = (ygrid.size, xgrid.size)
    nvar = 1 if values.ndim == 1 else values.shape[1]
    nsamples = values.shape[0]

    # trivial cases of the decorator (decorators.py:200-208)
    if nsamples == 1:
        out = np.ones((nvar,) + grid_shape)
        for i, v in enumerate(np.atleast_1d(values[0, ...])):
            out[i, ...] *= v
        return out.squeeze()
    if values.max() == values.min():
        return np.ones((nvar,) + grid_shape) * values.ravel()[0]

    ax, ay = _regular_axis(xgrid, "xgrid"), _regular_axis(ygrid, "ygrid")
    why = None
    if ax is None or ay is None or xy_coord.shape[1] != 2:
        why = "irregular grid or coordinates that are not 2-d"
    elif k is not None and 32 < k < nsamples:
        why = "k=%r (the kernel keeps at most 32 neighbours unless k covers all samples)" % (k,)
    elif not power > 0:
        why = "power=%r (the kernel needs power > 0)" % (power,)
    if why is not None:  # outside the kernel's limits: the reference takes the call
        ref = _reference_idw()
        if ref is None:
            raise NotImplementedError(
                "pysteps_amd idwinterp2d: %s is not implemented on the HIP path and pysteps is not importable" % why
            )
        warnings.warn("pysteps_amd idwinterp2d: %s -> delegating to the reference CPU path" % why)
        return ref(xy_coord, values, xgrid, ygrid, power=power, k=k, dist_offset=dist_offset, **kwargs)

    lib = _lib.lib()
    m, n = grid_shape
    cols = values.reshape(nsamples, nvar)
    k_eff = nsamples if k is None else int(min(k, nsamples))
    out = np.empty((nvar, m, n))
    xy64 = np.ascontiguousarray(xy_coord, dtype=np.float64)
    for c0 in range(0, nvar, 2):
        pair = np.zeros((nsamples, 2))
        width = min(2, nvar - c0)
        pair[:, :width] = cols[:, c0:c0 + width]
        buf = np.empty((2, m, n))
        rc = lib.psh_idw_host(
            xy64.ctypes.data, pair.ctypes.data, nsamples, m, n, ax[0], ax[1], ay[0], ay[1],
            k_eff, float(power), float(dist_offset), buf.ctypes.data,
        )
        _lib.check(rc, "psh_idw_host")
        out[c0:c0 + width] = buf[:width]
    return out.squeeze()
