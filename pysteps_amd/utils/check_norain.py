"""No-rain test, mirror of pysteps/utils/check_norain.py:6-58 (``check_norain``).

The callers either side of the advection path run it on the input field before they start
(nowcasts/steps.py:360, sprog.py:214, linda.py:304).  NumPy arrays are scanned on the host exactly
like the reference does; a :class:`~pysteps_amd.device.DeviceArray` is reduced where it lives
(``psh_count_above_dev``: the rain-pixel count and ``nanmin`` in two streaming passes) so that a
resident chain rain rate -> dB -> LK -> nowcast does not come back to the host for a yes/no answer.
"""

import ctypes

import numpy as np

from .. import _lib
from ..device import DeviceArray

__all__ = ["check_norain"]


def _window(m, n, win_fun):
    try:
        from pysteps.utils.tapering import compute_window_function  # noqa: PLC0415
    except Exception as exc:
        raise NotImplementedError(
            "check_norain: win_fun=%r needs pysteps.utils.tapering, which is not importable" % (win_fun,)) from exc
    return compute_window_function(m, n, win_fun)


def check_norain(precip_arr, precip_thr=None, norain_thr=0.0, win_fun=None, printmsg=True):
    """True if the fraction of pixels above ``precip_thr`` (default: the field minimum) is at most
    ``norain_thr``.  Parameters and return value as the reference (:9-33)."""
    if isinstance(precip_arr, DeviceArray) and win_fun is None:
        if precip_arr.dtype != np.float32:
            raise ValueError("device-resident fields must be float32")
        count, lowest = ctypes.c_double(), ctypes.c_double()
        # `masked > precip_thr` (check_norain.py:49) on a float32 field: NumPy rounds a Python float (and a
        # numpy.float32) to float32 first and compares a numpy.float64 scalar in float64; the kernel
        # compares in double, so the rounding NumPy would apply is applied here
        if precip_thr is None:
            thr = float("nan")
        elif isinstance(precip_thr, np.floating) and np.dtype(type(precip_thr)).itemsize > 4:
            thr = float(precip_thr)
        else:
            thr = float(np.float32(precip_thr))
        _lib.check(_lib.lib().psh_count_above_dev(precip_arr.ptr, precip_arr.size, thr, ctypes.byref(count),
                                                  ctypes.byref(lowest)), "psh_count_above_dev")
        fraction = count.value / precip_arr.size
    else:
        arr = precip_arr.to_host() if isinstance(precip_arr, DeviceArray) else np.asarray(precip_arr)
        masked = arr.copy()
        if win_fun is not None:
            masked[..., _window(arr.shape[-2], arr.shape[-1], win_fun) == 0.0] = np.nanmin(arr)
        if precip_thr is None:
            precip_thr = np.nanmin(masked)
        fraction = np.count_nonzero(masked > precip_thr) / masked.size
    if printmsg:
        print(f"Rain fraction is: {str(fraction)}, while minimum fraction is {str(norain_thr)}")
    return fraction <= norain_thr
