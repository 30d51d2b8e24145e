"""FFT-based scale decomposition on the GPU (mirror of pysteps/cascade/decomposition.py).

``decomposition_fft`` / ``recompose_fft`` with the reference's signatures, dictionaries and error
behaviour (pysteps/cascade/decomposition.py:77-305).  The form the STEPS member loop uses - spatial
input, spatial output, no mask, a field whose sides are powers of two - runs as one device pipeline
(csrc/cascade.hip: one forward transform, per level an inverse transform with the band-pass weights
applied inside its column pass, level statistics, normalisation); every other combination of options
is the reference's own code path with the HIP FFT method object (``pysteps_amd.utils.fft``) or, for
shapes the kernels do not take, plain ``numpy.fft`` - same results either way.

The band-pass weights of a filter (``bp_filter["weights_2d"]``, levels x m x n/2+1 float64: 0.5 GB
at 4096 x 4096 x 8) are uploaded once and kept on the device for as long as the filter's array lives.
"""

import ctypes
import weakref

import numpy as np

from .. import _lib
from ..device import DeviceArray
from ..utils import fft as hip_fft

_weights_cache = {}  # id(host array) -> (weakref, fingerprint, device copy, nbytes), oldest first
WEIGHTS_CACHE_BYTES = 4 << 30  # device copies kept at most (a 4096^2 x 8-level filter is 0.5 GB)
MIN_HOST_PLANE = 1 << 16  # host cascades below this many pixels per level are summed by NumPy


def _fingerprint(arr):
    """Cheap content check of a cached array: shape, dtype, sum of a strided sample, corner values -
    an in-place edit of the host array must not keep a stale device copy in use."""
    flat = arr.reshape(-1)
    step = max(1, flat.size // 4099)
    return (arr.shape, arr.dtype.str, float(np.sum(flat[::step], dtype=np.float64)), float(flat[0]), float(flat[-1]))


def _device_weights(weights):
    """Device copy of a (levels, m, n/2+1) weight array, cached by the identity of the host array and
    a fingerprint of its content; the cache holds at most WEIGHTS_CACHE_BYTES (oldest copy dropped)."""
    key = id(weights)
    hit = _weights_cache.get(key)
    fp = _fingerprint(np.asarray(weights))
    if hit is not None and hit[0]() is weights and hit[1] == fp:
        return hit[2]
    dev = DeviceArray.from_host(np.ascontiguousarray(weights, dtype=np.float64))
    try:
        ref = weakref.ref(weights, lambda _r, k=key: _weights_cache.pop(k, None))
    except TypeError:  # not weak-referenceable: do not cache
        return dev
    _weights_cache.pop(key, None)
    _weights_cache[key] = (ref, fp, dev, dev.nbytes)
    while len(_weights_cache) > 1 and sum(v[3] for v in _weights_cache.values()) > WEIGHTS_CACHE_BYTES:
        _weights_cache.pop(next(iter(_weights_cache)))
    return dev


def invalidate_weights_cache():
    """Drop every cached device copy of band-pass weights / noise filters."""
    _weights_cache.clear()


def _device_nonfinite(field):
    """True if a float64 DeviceArray holds a NaN or an infinity (one reduction pass on the device)."""
    count = ctypes.c_double(0.0)
    _lib.check(_lib.lib().psh_nonfinite_count_f64_dev(field.ptr, field.size, ctypes.byref(count)), "psh_nonfinite_count_f64_dev")
    return count.value > 0


def _reference_decomposition():
    try:
        from pysteps.cascade.decomposition import decomposition_fft as ref  # noqa: PLC0415
    except Exception:
        return None
    return None if ref is decomposition_fft else ref


def _reference_recompose():
    try:
        from pysteps.cascade.decomposition import recompose_fft as ref  # noqa: PLC0415
    except Exception:
        return None
    return None if ref is recompose_fft else ref


def decomposition_fft(field, bp_filter, **kwargs):
    """Decompose a two-dimensional field into multiple spatial scales (reference:
    pysteps/cascade/decomposition.py:77-262; parameters, defaults and the returned dictionary as
    documented there)."""
    normalize = kwargs.get("normalize", False)
    mask = kwargs.get("mask", None)
    input_domain = kwargs.get("input_domain", "spatial")
    output_domain = kwargs.get("output_domain", "spatial")
    compute_stats = kwargs.get("compute_stats", True)
    subtract_mean = kwargs.get("subtract_mean", False)
    if normalize and not compute_stats:
        compute_stats = True

    resident = isinstance(field, DeviceArray)
    shape = tuple(field.shape)
    on_device = (
        len(shape) == 2 and hip_fft.supported_shape(shape) and mask is None and input_domain == "spatial"
        and output_domain == "spatial" and (resident or np.asarray(field).dtype != np.float32)
    )
    if not on_device:
        ref = _reference_decomposition()
        if ref is None or resident:
            raise NotImplementedError(
                "pysteps_amd decomposition_fft: only spatial -> spatial, unmasked, power-of-two fields "
                "run on the HIP path and pysteps is not importable for the rest"
            )
        kw = dict(kwargs)
        fft = kw.get("fft_method", "numpy")
        if isinstance(fft, str) and len(shape) == 2:  # the reference's transforms, through the HIP method object
            kw["fft_method"] = hip_fft.get_hip(shape) if hip_fft.supported_shape(shape) else fft
        return ref(field, bp_filter, **kw)

    # ---- checks of the reference (decomposition.py:148-196), same messages -----------------------
    weights = bp_filter["weights_2d"]
    if shape[0] != weights.shape[1]:
        raise ValueError(
            "dimension mismatch between field and bp_filter: "
            + "field.shape[0]=%d , " % shape[0]
            + "bp_filter['weights_2d'].shape[1]"
            "=%d" % weights.shape[1]
        )
    if int(shape[1] / 2) + 1 != weights.shape[2]:
        raise ValueError(
            "Dimension mismatch between field and bp_filter: "
            "int(field.shape[1]/2)+1=%d , " % (int(shape[1] / 2) + 1)
            + "bp_filter['weights_2d'].shape[2]"
            "=%d" % weights.shape[2]
        )
    if not resident and np.any(~np.isfinite(field)):
        raise ValueError("field contains non-finite values")
    if resident and field.dtype == np.float64 and _device_nonfinite(field):  # decomposition.py:195-196, on the device
        raise ValueError("field contains non-finite values")

    m, n = shape
    nlevels = len(bp_filter["weights_1d"])
    if weights.shape[0] != nlevels:  # the reference would fail while indexing weights_2d[k]
        raise ValueError("dimension mismatch inside bp_filter: len(weights_1d)=%d, weights_2d.shape[0]=%d"
                         % (nlevels, weights.shape[0]))
    d_field = field if resident else DeviceArray.from_host(np.ascontiguousarray(field, dtype=np.float64))
    if d_field.dtype != np.float64:
        raise ValueError("device-resident fields must be float64")
    d_weights = _device_weights(weights)
    levels = DeviceArray((nlevels, m, n), np.float64)
    means = (ctypes.c_double * nlevels)()
    stds = (ctypes.c_double * nlevels)()
    field_mean = ctypes.c_double(0.0)
    _lib.check(
        _lib.lib().psh_cascade_decompose_dev(d_field.ptr, d_weights.ptr, nlevels, m, n, 1 if normalize else 0,
                                             1 if subtract_mean else 0, levels.ptr, means, stds,
                                             ctypes.byref(field_mean)),
        "psh_cascade_decompose_dev",
    )
    result = {}
    if subtract_mean:
        result["field_mean"] = np.float64(field_mean.value)
    result["domain"] = output_domain
    result["normalized"] = normalize
    result["compact_output"] = False
    result["cascade_levels"] = levels if resident else levels.to_host()
    if compute_stats:
        result["means"] = [np.float64(v) for v in means]
        result["stds"] = [np.float64(v) for v in stds]
    return result


def recompose_fft(decomp, **kwargs):
    """Recompose a cascade obtained with decomposition_fft (reference:
    pysteps/cascade/decomposition.py:265-305).  Device-resident cascades (``cascade_levels`` a
    DeviceArray) are summed on the GPU, and so are float64 host cascades of spatial decompositions
    from 65536 pixels per level on; other host cascades take the reference's NumPy expression."""
    levels = decomp["cascade_levels"]
    if (
        isinstance(levels, np.ndarray) and levels.ndim == 3 and levels.dtype == np.float64
        and levels.shape[1] * levels.shape[2] >= MIN_HOST_PLANE and 1 <= levels.shape[0] <= 64
        and decomp["domain"] == "spatial" and not decomp.get("compact_output", False)
    ):
        # host cascade of a spatial decomposition: the same sum on the device, bit-identical with the
        # NumPy expression below (csrc/cascade.hip `recompose`), one transfer of the levels instead of
        # nlevels + 2 full-size temporaries
        on_device = dict(decomp)
        on_device["cascade_levels"] = DeviceArray.from_host(levels, sync=False)
        return recompose_fft(on_device).to_host()
    if not isinstance(levels, DeviceArray):
        # everything else (small cascades, spectral / compact ones): the reference's own function
        ref = _reference_recompose()
        if ref is None:
            raise NotImplementedError("pysteps_amd recompose_fft: this cascade is not taken by the HIP path and pysteps is not "
                                      "importable for the reference's recompose_fft")
        return ref(decomp, **kwargs)

    if decomp["domain"] != "spatial":
        raise NotImplementedError("pysteps_amd recompose_fft: device-resident cascades are spatial")
    nlevels, m, n = levels.shape
    out = DeviceArray((m, n), np.float64)
    if decomp["normalized"]:
        mu = (ctypes.c_double * nlevels)(*[float(v) for v in decomp["means"]])
        sigma = (ctypes.c_double * nlevels)(*[float(v) for v in decomp["stds"]])
    else:
        mu = sigma = None
    _lib.check(
        _lib.lib().psh_cascade_recompose_dev(levels.ptr, nlevels, m, n, mu, sigma,
                                             float(decomp.get("field_mean", 0.0)), out.ptr),
        "psh_cascade_recompose_dev",
    )
    return out
