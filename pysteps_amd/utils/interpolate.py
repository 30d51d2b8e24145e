"""Sparse -> dense interpolation on MI355X, mirror of pysteps/utils/interpolate.py.

``idwinterp2d`` has the reference's signature and return convention
(interpolate.py:26-114) including the input checks and trivial cases its
``prepare_interpolator`` decorator performs (decorators.py:153-250); the k-NN search
and the weighting run in the HIP kernel ``csrc/idw.hip`` through ``psh_idw_*``.

``rbfinterp2d`` (interpolate.py:117-170, a wrapper of ``scipy.interpolate.Rbf``): the weights
are SciPy's own dense solve on the host (N x N, N = number of sparse vectors), the evaluation
of the interpolant at every grid node - where the reference spends its time - runs in
``csrc/rbf.hip`` through ``psh_rbf_eval_dev``.
"""

import ctypes
import warnings

import numpy as np

from .. import _lib
from ..device import DeviceArray

__all__ = ["idwinterp2d", "rbfinterp2d"]


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


_RBF_FUNCTIONS = {"multiquadric": 0, "inverse": 1, "gaussian": 2, "linear": 3, "cubic": 4, "quintic": 5, "thin_plate": 6}


def _reference_rbf():
    try:
        from pysteps.utils.interpolate import rbfinterp2d as ref  # noqa: PLC0415
    except Exception:
        return None
    return None if ref is rbfinterp2d else ref


def rbfinterp2d(xy_coord, values, xgrid, ygrid, **kwargs):
    """Radial basis function interpolation of a sparse (multivariate) array.

    Parameters and return value as in the reference (interpolate.py:117-170 with the preamble of
    decorators.py:153-250): ``(ygrid.size, xgrid.size)`` for 1-d ``values`` or ``(m, ygrid.size,
    xgrid.size)`` float64; keyword arguments are those of ``scipy.interpolate.Rbf`` (``function``,
    ``epsilon``, ``smooth``; ``nchunks`` / ``hkey`` are accepted - the result does not depend on the
    chunking the reference uses to bound memory).  A callable ``function``, a ``norm`` other than the
    Euclidean one, coordinates that are not 2-d or an irregular grid go to the reference.
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

    deprecated = [arg for arg in ("rbfunction", "k") if arg in kwargs]
    if deprecated:  # interpolate.py:151-159
        warnings.warn("rbfinterp2d: The following keyword arguments are deprecated:\n" + str(deprecated), DeprecationWarning)
    rbf_kwargs = {k: v for k, v in kwargs.items() if k not in ("nchunks", "hkey")}
    function = rbf_kwargs.get("function", "multiquadric")
    ax, ay = _regular_axis(xgrid, "xgrid"), _regular_axis(ygrid, "ygrid")
    why = None
    if ax is None or ay is None or xy_coord.shape[1] != 2:
        why = "irregular grid or coordinates that are not 2-d"
    elif values.ndim == 1 and _reference_rbf() is not None:
        # (the reference moves the LAST axis of the Rbf result to the front, interpolate.py:169 - for 1-d values
        # that is a transposition, per grid chunk, and an error on non-square chunks: its behaviour there is its own)
        why = "1-d values (the reference's own axis handling applies)"
    elif not isinstance(function, str) or function not in _RBF_FUNCTIONS:
        why = "function=%r (the kernel knows %s)" % (function, ", ".join(sorted(_RBF_FUNCTIONS)))
    elif "norm" in rbf_kwargs and rbf_kwargs["norm"] not in (None, "euclidean"):
        why = "norm=%r (the kernel evaluates Euclidean distances)" % (rbf_kwargs["norm"],)
    elif any(k not in ("function", "epsilon", "smooth", "norm", "mode", "rbfunction", "k") for k in rbf_kwargs):
        why = "keyword arguments %r" % sorted(rbf_kwargs)
    if why is not None:  # outside the kernel's limits: the reference takes the call
        ref = _reference_rbf()
        if ref is None:
            raise NotImplementedError(
                "pysteps_amd rbfinterp2d: %s is not implemented on the HIP path and pysteps is not importable" % why
            )
        warnings.warn("pysteps_amd rbfinterp2d: %s -> delegating to the reference CPU path" % why)
        return ref(xy_coord, values, xgrid, ygrid, **kwargs)

    # the weights: SciPy's own solve, with the keyword arguments the reference would hand it (interpolate.py:161-167)
    from scipy.interpolate import Rbf  # noqa: PLC0415

    solve_kwargs = {k: v for k, v in rbf_kwargs.items() if k in ("function", "epsilon", "smooth", "norm", "rbfunction", "k")}
    solve_kwargs["mode"] = "1-D" if values.ndim == 1 else "N-D"
    rbfi = Rbf(*np.split(xy_coord, xy_coord.shape[1], 1), values, **solve_kwargs)
    nodes = np.asarray(rbfi.nodes, dtype=np.float64).reshape(nsamples, nvar)
    epsilon = float(rbfi.epsilon) if rbfi.epsilon is not None else 1.0

    lib = _lib.lib()
    m, n = grid_shape
    xy_d = DeviceArray.from_host(np.ascontiguousarray(xy_coord, dtype=np.float64))
    out = np.empty((nvar, m, n))
    buf = DeviceArray((2, m, n), np.float64)
    for c0 in range(0, nvar, 2):
        width = min(2, nvar - c0)
        pair = np.zeros((nsamples, 2))
        pair[:, :width] = nodes[:, c0:c0 + width]
        w_d = DeviceArray.from_host(pair)
        _lib.check(
            lib.psh_rbf_eval_dev(xy_d.ptr, w_d.ptr, nsamples, m, n, ax[0], ax[1], ay[0], ay[1],
                                 _RBF_FUNCTIONS[function], epsilon, buf.ptr),
            "psh_rbf_eval_dev",
        )
        out[c0:c0 + width] = buf.to_host()[:width]
    return out.squeeze()
