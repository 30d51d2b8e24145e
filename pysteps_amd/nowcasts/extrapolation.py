"""Deterministic Lagrangian-persistence nowcast, mirror of pysteps/nowcasts/extrapolation.py:19-117.

One extrapolator call with the reference's input checks; resolves ``extrap_method`` through
:func:`pysteps_amd.extrapolation.get_method`, so "semilagrangian" is the HIP kernel.  With
DeviceArray inputs the non-finite scan of the reference (:76) is a device reduction and the
forecast stays in HBM.
"""

import time

import numpy as np

from .. import extrapolation
from ..device import DeviceArray

__all__ = ["forecast"]


def _check_inputs(precip, velocity, timesteps):
    if precip.ndim != 2:
        raise ValueError("The input precipitation must be a two-dimensional array")
    if velocity.ndim != 3:
        raise ValueError("Input velocity must be a three-dimensional array")
    if tuple(precip.shape) != tuple(velocity.shape[1:3]):
        raise ValueError(
            "Dimension mismatch between input precipitation and velocity: "
            "shape(precip)=%s, shape(velocity)=%s" % (str(tuple(precip.shape)), str(tuple(velocity.shape)))
        )
    if isinstance(timesteps, list) and not sorted(timesteps) == timesteps:
        raise ValueError("timesteps is not in ascending order")


def forecast(precip, velocity, timesteps, extrap_method="semilagrangian", extrap_kwargs=None,
             measure_time=False):
    """Generate a nowcast by applying a simple advection-based extrapolation to the given
    precipitation field.  Parameters and returns as in the reference (:27-66): array
    ``(num_timesteps, m, n)``, or ``(array, seconds)`` with ``measure_time=True``."""
    _check_inputs(precip, velocity, timesteps)
    extrap_kwargs = dict() if extrap_kwargs is None else extrap_kwargs.copy()
    if isinstance(precip, DeviceArray):
        from ..utils.transformation import field_stats  # noqa: PLC0415

        extrap_kwargs["allow_nonfinite_values"] = field_stats(precip)[2] > 0
    else:
        extrap_kwargs["allow_nonfinite_values"] = bool(np.any(~np.isfinite(precip)))
    if measure_time:
        print(f"Computing extrapolation nowcast from a {precip.shape[0]:d}x{precip.shape[1]:d} input grid... ", end="")
        start_time = time.time()
    method = extrapolation.get_method(extrap_method)
    precip_forecast = method(precip, velocity, timesteps, **extrap_kwargs)
    if measure_time:
        if isinstance(precip_forecast, DeviceArray):
            from ..device import synchronize  # noqa: PLC0415

            synchronize()
        computation_time = time.time() - start_time
        print(f"{computation_time:.2f} seconds.")
        return precip_forecast, computation_time
    return precip_forecast
