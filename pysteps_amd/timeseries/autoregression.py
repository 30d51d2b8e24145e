"""One step of the AR(p) model of a cascade level on the GPU (mirror of
pysteps/timeseries/autoregression.py:1020-1070, ``iterate_ar_model``).

The member loops call it once per member, cascade level and time step (pysteps/nowcasts/steps.py:1095,
1137, sprog.py:398, sseps.py:678,749, anvil.py:483).  ``psh_ar_iterate_dev`` evaluates the reference's
expression ``x_new = 0.0; x_new += phi[i] * x[-(i + 1)]; x_new += phi[-1] * eps`` with every product
rounded before it is added, so float64 results are bit-identical with NumPy's.

Device path: float64 series of at least two dimensions, scalar ``phi`` of order 1..8, ``eps`` of
``x.shape[1:]`` or None; ``DeviceArray`` in gives ``DeviceArray`` out (the form a resident member loop
uses).  Everything else - one-dimensional series, other dtypes, per-pixel ``phi``, small host arrays
(the transfer would cost more than NumPy's arithmetic) - runs the reference's function.
"""

import ctypes

import numpy as np

from .. import _lib
from ..device import DeviceArray

MAX_ORDER = 8
MIN_HOST_PLANE = 1 << 16  # host arrays below this many values per field stay with NumPy

_reference_fn = None  # set by register.patch_autoregression(): the function this module replaced


def _reference():
    if _reference_fn is not None:
        return _reference_fn
    from pysteps.timeseries import autoregression as ref_mod  # noqa: PLC0415

    fn = getattr(ref_mod, "_reference_iterate_ar_model", ref_mod.iterate_ar_model)
    if fn is iterate_ar_model:
        raise NotImplementedError("the reference's iterate_ar_model is not reachable")
    return fn


def _scalar_phi(phi):
    try:
        if not 2 <= len(phi) <= MAX_ORDER + 1:
            return None
        if any(np.ndim(v) != 0 for v in phi):
            return None
        return np.asarray(phi, dtype=np.float64)
    except TypeError:
        return None


def iterate_ar_model(x, phi, eps=None):
    """Apply an AR(p) model to a time series (parameters and return value as documented for the
    reference, autoregression.py:1021-1040)."""
    resident = isinstance(x, DeviceArray)
    coeffs = _scalar_phi(phi)
    eligible = (
        coeffs is not None and len(x.shape) >= 2 and x.dtype == np.float64
        and (eps is None or (isinstance(eps, (np.ndarray, DeviceArray)) and eps.dtype == np.float64))
    )
    if resident:
        if not eligible or (eps is not None and not isinstance(eps, DeviceArray)):
            raise NotImplementedError("device-resident series: float64, scalar phi of order 1..8, resident eps")
    elif not eligible or isinstance(eps, DeviceArray) or int(np.prod(x.shape[1:])) < MIN_HOST_PLANE:
        return _reference()(x, phi, eps=eps)
    if x.shape[0] < len(phi) - 1:  # autoregression.py:1041-1045
        raise ValueError(
            "dimension mismatch between x and phi: x.shape[0]=%d, len(phi)=%d" % (x.shape[0], len(phi))
        )
    if eps is not None and tuple(eps.shape) != tuple(x.shape[1:]):  # :1053-1057
        raise ValueError(
            "dimension mismatch between x and eps: x[1:].shape=%s, eps.shape=%s"
            % (str(tuple(x.shape[1:]) if resident else x[1:].shape), str(tuple(eps.shape)))
        )
    plane = int(np.prod(x.shape[1:]))
    if plane == 0:
        return _reference()(x, phi, eps=eps)
    d_x = x if resident else DeviceArray.from_host(x, np.float64, sync=False)
    d_eps = eps if resident or eps is None else DeviceArray.from_host(eps, np.float64, sync=False)
    out = DeviceArray(x.shape, np.float64)
    phi_c = (ctypes.c_double * coeffs.size)(*coeffs)
    _lib.check(
        _lib.lib().psh_ar_iterate_dev(d_x.ptr, int(x.shape[0]), plane, phi_c, coeffs.size - 1,
                                      None if d_eps is None else d_eps.ptr, out.ptr),
        "psh_ar_iterate_dev",
    )
    return out if resident else out.to_host()  # blocks are recycled in stream order: inputs may die now
