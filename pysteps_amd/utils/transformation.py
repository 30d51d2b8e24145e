"""dB transform on the device, mirror of pysteps/utils/transformation.py:150-232 (``dB_transform``).

NumPy inputs are transformed on the host exactly like the reference does (it is one
element-wise pass); :class:`~pysteps_amd.device.DeviceArray` inputs stay in HBM
(``psh_db_transform_dev``) so that rain rate -> dB -> LK -> extrapolation -> rain rate
never crosses PCIe.
"""

import ctypes

import numpy as np

from .. import _lib
from ..device import DeviceArray

__all__ = ["dB_transform", "field_stats"]


def field_stats(field):
    """(min, max, number of non-finite values) of a float32 DeviceArray, computed on the device."""
    mn, mx, bad = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _lib.check(
        _lib.lib().psh_field_stats_dev(field.ptr, field.size, ctypes.byref(mn), ctypes.byref(mx), ctypes.byref(bad)),
        "psh_field_stats_dev",
    )
    return mn.value, mx.value, int(bad.value)


def dB_transform(R, metadata=None, threshold=None, zerovalue=None, inverse=False):
    """Methods to transform precipitation intensities to/from dB units.

    Same parameters, return value ``(R, metadata)`` and metadata updates as the
    reference (transformation.py:150-232).
    """
    if metadata is None:
        metadata = {"transform": "dB"} if inverse else {"transform": None}
    else:
        metadata = metadata.copy()
    on_device = isinstance(R, DeviceArray)

    if not inverse:
        if metadata["transform"] == "dB":
            return (R if on_device else R.copy()), metadata
        if threshold is None:
            threshold = metadata.get("threshold", 0.1)
        threshold_db = 10.0 * np.log10(threshold)
        if zerovalue is None:
            zerovalue = threshold_db - 5
        if on_device:
            out = DeviceArray(R.shape, np.float32)
            _lib.check(_lib.lib().psh_db_transform_dev(R.ptr, out.ptr, R.size, float(threshold), float(zerovalue), 0),
                       "psh_db_transform_dev")
        else:
            out = R.copy()
            zeros = out < threshold
            out[~zeros] = 10.0 * np.log10(out[~zeros])
            out[zeros] = zerovalue
        metadata["transform"] = "dB"
        metadata["zerovalue"] = zerovalue
        metadata["threshold"] = threshold_db
        return out, metadata

    if metadata["transform"] != "dB":
        return (R if on_device else R.copy()), metadata
    if threshold is None:
        threshold = metadata.get("threshold", -10.0)
    if zerovalue is None:
        zerovalue = 0.0
    threshold_lin = 10.0 ** (threshold / 10.0)
    if on_device:
        out = DeviceArray(R.shape, np.float32)
        _lib.check(_lib.lib().psh_db_transform_dev(R.ptr, out.ptr, R.size, float(threshold), float(zerovalue), 1),
                   "psh_db_transform_dev")
    else:
        out = 10.0 ** (R / 10.0)
        out[out < threshold_lin] = zerovalue
    metadata["transform"] = None
    metadata["threshold"] = threshold_lin
    metadata["zerovalue"] = zerovalue
    return out, metadata
