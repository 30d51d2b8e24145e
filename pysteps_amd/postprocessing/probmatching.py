"""Empirical-CDF probability matching on the GPU (mirror of
pysteps/postprocessing/probmatching.py:55-140, ``nonparam_match_empirical_cdf``).

The member loops call it once per member and time step with the recomposed forecast and the latest
observation (pysteps/nowcasts/steps.py:1199, sprog.py:421, sseps.py:783,804); in the reference it is
two ``argsort`` calls over the whole grid.  csrc/probmatch.hip ranks the wet pixels of both arrays
with one bucket pass each and writes the matched field directly (``psh_probmatch_dev``).

What runs on the device is the call form of those loops: ``ignore_indices=None``.  Calls with
``ignore_indices``, and inputs the device path declines (more than 16384 wet values of the initial
array tied or crowded into one of its 2**20 value buckets, infinities in the target), go to the
reference's function, and so do host arrays below 4096 pixels (the localised windows of
nowcasts/sseps.py:783: two NumPy sorts of that size take less than the launch chain).  Tied wet values of the initial array are ranked in pixel order (NumPy's stable
sort); the reference's quicksort leaves their order unspecified, so fields with tied wet values agree
with the reference as multisets per tie group, everything else bit for bit.
"""

import numpy as np

from .. import _lib
from ..device import DeviceArray

MIN_HOST_SIZE = 4096  # host arrays below this many pixels: two NumPy sorts beat the launch chain + transfers

_reference_fn = None  # set by register.patch_probmatching(): the function this module replaced


def _reference():
    if _reference_fn is not None:
        return _reference_fn
    from pysteps.postprocessing import probmatching as ref_mod  # noqa: PLC0415

    fn = getattr(ref_mod, "_reference_nonparam_match_empirical_cdf", ref_mod.nonparam_match_empirical_cdf)
    if fn is nonparam_match_empirical_cdf:
        raise NotImplementedError("the reference's nonparam_match_empirical_cdf is not reachable")
    return fn


def nonparam_match_empirical_cdf(initial_array, target_array, ignore_indices=None):
    """Matches the empirical CDF of the initial array with the empirical CDF of a target array
    (parameters and return value as documented for the reference, probmatching.py:56-79).
    ``DeviceArray`` inputs (float64) give a ``DeviceArray``."""
    resident = isinstance(initial_array, DeviceArray) and isinstance(target_array, DeviceArray)
    if ignore_indices is not None:
        if resident or isinstance(initial_array, DeviceArray) or isinstance(target_array, DeviceArray):
            raise NotImplementedError("ignore_indices is not available for device-resident arrays")
        return _reference()(initial_array, target_array, ignore_indices=ignore_indices)
    if not resident:
        if isinstance(initial_array, DeviceArray):
            initial_array = initial_array.to_host()
        if isinstance(target_array, DeviceArray):
            target_array = target_array.to_host()
        initial_array = np.asarray(initial_array)
        target_array = np.asarray(target_array)
    if initial_array.size != target_array.size:
        raise ValueError(
            "dimension mismatch between initial_array and target_array: "
            f"initial_array.shape={initial_array.shape}, target_array.shape={target_array.shape}"
        )
    if initial_array.size == 0:
        return _reference()(initial_array, target_array)  # numpy's own error for empty reductions
    if not resident and initial_array.size < MIN_HOST_SIZE:
        try:  # small windows (nowcasts/sseps.py:783): the reference is faster; without pysteps, the device
            return _reference()(initial_array, target_array)
        except (ImportError, NotImplementedError):
            pass
    if resident:
        if initial_array.dtype != np.float64 or target_array.dtype != np.float64:
            raise ValueError("device-resident arrays must be float64")
        d_init, d_trg = initial_array, target_array
    else:
        d_init = DeviceArray.from_host(initial_array, np.float64, sync=False)
        d_trg = DeviceArray.from_host(target_array, np.float64, sync=False)
    out = DeviceArray(initial_array.shape, np.float64)
    rc = _lib.lib().psh_probmatch_dev(d_init.ptr, d_trg.ptr, initial_array.size, out.ptr)
    if rc == _lib.PSH_EUNSUPPORTED and not resident:
        return _reference()(initial_array, target_array)
    if rc == _lib.PSH_EUNSUPPORTED:
        # resident arrays the bucket pass declines (more than 16384 tied or bucket-sharing wet values,
        # infinities in the target): this one call crosses the bus, the member loop keeps going
        got = _reference()(initial_array.to_host(), target_array.to_host())
        return DeviceArray.from_host(np.ascontiguousarray(got, dtype=np.float64))
    _lib.check(rc, "psh_probmatch_dev")
    return out if resident else out.to_host()
