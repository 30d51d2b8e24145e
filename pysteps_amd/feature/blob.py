"""Scale-space blob detection on MI355X, drop-in for ``pysteps.feature.blob.detection``
(reference: pysteps/feature/blob.py:32-140).

The reference hands the image to scikit-image's ``blob_log`` / ``blob_dog`` (third party; 0.18.3:
``skimage/feature/blob.py``, ``peak.py``), i.e. to ``scipy.ndimage.gaussian_laplace`` / ``gaussian_filter`` once per
scale, a 3 x 3 x 3 ``maximum_filter`` over the scale cube and a pairwise pruning of overlapping blobs.  Here the cube,
the maximum filter and the peak mask are HIP kernels (``csrc/blob.hip``: SciPy's arithmetic operation by operation,
NaN semantics of its ring-buffer maximum filter included); what works on the few hundred peaks - the ordering by
response, scikit-image's ``_prune_blobs`` (restated below, ``scipy.spatial.cKDTree`` like the original), the
``max_num_features`` selection - stays on the host.  No scikit-image needed.

Same signature, return values and exceptions.  ``method="doh"`` (determinant of Hessian on integral images, a Cython
routine of scikit-image) is not implemented: forwarded to the reference when pysteps is importable.
"""

import ctypes
import math
import warnings

import numpy as np
from scipy import spatial

from .. import _lib
from ..device import DeviceArray

__all__ = ["detection", "response_cube"]


def _gaussian_kernel1d(sigma, order, radius):
    """scipy/ndimage/_filters.py ``_gaussian_kernel1d`` (the weights ``gaussian_filter1d`` correlates with), restated:
    the same NumPy expressions, so the same doubles."""
    exponent_range = np.arange(order + 1)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    if order == 0:
        return phi_x
    q = np.zeros(order + 1)
    q[0] = 1
    D = np.diag(exponent_range[1:], 1)
    P = np.diag(np.ones(order) / -sigma2, -1)
    Q_deriv = D + P
    for _ in range(order):
        q = Q_deriv.dot(q)
    q = (x[:, None] ** exponent_range).dot(q)
    return q * phi_x


def _half_kernels(sigmas, truncate=4.0):
    """Per scale: the radius ``int(truncate * sigma + 0.5)`` and [centre, 1 .. radius] of the smoothing and of the
    second-derivative kernel (correlate1d's symmetric branch reads the centre and the LEFT half of the reversed
    weights: element ``radius - j`` of ``weights[::-1]``)."""
    radii, parts = [], []
    for s in sigmas:
        sd = float(s)
        lw = int(truncate * sd + 0.5)
        for order in (0, 2):
            w = _gaussian_kernel1d(sd, order, lw)[::-1]
            parts.append(np.ascontiguousarray(w[lw::-1]))  # w[lw], w[lw - 1], .., w[0]: centre, then distance 1 .. lw
        radii.append(lw)
    return np.asarray(radii, dtype=np.int32), np.concatenate(parts).astype(np.float64)


def _sigma_list(method, min_sigma, max_sigma, kwargs):
    if method == "log":  # skimage/feature/blob.py blob_log
        num_sigma = kwargs.get("num_sigma", 10)
        if kwargs.get("log_scale", False):
            return np.logspace(np.log10(float(min_sigma)), np.log10(float(max_sigma)), num_sigma)
        return np.linspace(0, 1, num_sigma) * (float(max_sigma) - float(min_sigma)) + float(min_sigma)
    ratio = kwargs.get("sigma_ratio", 1.6)  # blob_dog
    k = int(np.mean(np.log(float(max_sigma) / float(min_sigma)) / np.log(ratio) + 1))
    return np.array([float(min_sigma) * (ratio ** i) for i in range(k + 1)])


def _to_device(image):
    if isinstance(image, DeviceArray):
        if image.dtype not in (np.float32, np.float64):
            raise ValueError("device-resident input_image must be float32 or float64")
        return image
    arr = np.asarray(image)  # (a MaskedArray's data, like skimage's img_as_float: masked pixels keep what they hold)
    if arr.dtype not in (np.float32, np.float64):
        raise NotImplementedError("pysteps_amd blob.detection: float32 / float64 images (img_as_float rescales integer ones)")
    return DeviceArray.from_host(np.ascontiguousarray(arr), dtype=arr.dtype)


def response_cube(image_dev, sigmas, method="log"):
    """(K, m, n) float64 DeviceArray of scale-normalised responses: ``-gaussian_laplace(image, s) * s**2`` per scale
    (``method="log"``, K = len(sigmas)) or ``(G(s_k) - G(s_k+1)) * s_k`` (``"dog"``, K = len(sigmas) - 1)."""
    m, n = image_dev.shape
    sigmas = np.ascontiguousarray(sigmas, dtype=np.float64)
    radii, weights = _half_kernels(sigmas)
    K = len(sigmas) if method == "log" else len(sigmas) - 1
    cube = DeviceArray((K, m, n), np.float64)
    _lib.check(
        _lib.lib().psh_blob_cube_dev(image_dev.ptr, 1 if image_dev.dtype == np.float32 else 0, m, n, 0 if method == "log" else 1,
                                     sigmas.ctypes.data, len(sigmas), radii.ctypes.data, weights.ctypes.data, cube.ptr),
        "psh_blob_cube_dev",
    )
    return cube


def _peaks(cube, threshold):
    """peak_local_max(cube, threshold_abs=threshold, footprint=ones(3,3,3), threshold_rel=0, exclude_border=False)
    -> ((p, 3) int (row, col, scale index), (p,) values), strongest first (peak.py ``_get_high_intensity_peaks``:
    coordinates in C order of the (m, n, K) cube, then ``argsort(-values)``; exact ties keep that order here)."""
    K, m, n = cube.shape
    lib = _lib.lib()
    capacity = 1 << 16
    while True:
        coords = np.empty((capacity, 3), dtype=np.int32)
        values = np.empty(capacity, dtype=np.float64)
        count = ctypes.c_int(0)
        _lib.check(lib.psh_blob_peaks_dev(cube.ptr, K, m, n, float(threshold), capacity, coords.ctypes.data, values.ctypes.data,
                                          ctypes.byref(count)), "psh_blob_peaks_dev")
        if count.value <= capacity:
            break
        capacity = count.value
    coords, values = coords[: count.value], values[: count.value]
    order = np.lexsort((coords[:, 2], coords[:, 1], coords[:, 0]))  # np.nonzero's order
    coords, values = coords[order], values[order]
    idx = np.argsort(-values, kind="stable")
    return coords[idx], values[idx]


def _disk_overlap(d, r1, r2):  # skimage/feature/blob.py _compute_disk_overlap
    ratio1 = min(max((d ** 2 + r1 ** 2 - r2 ** 2) / (2 * d * r1), -1), 1)
    ratio2 = min(max((d ** 2 + r2 ** 2 - r1 ** 2) / (2 * d * r2), -1), 1)
    a, b, c, e = -d + r2 + r1, d - r2 + r1, d + r2 - r1, d + r2 + r1
    area = r1 ** 2 * math.acos(ratio1) + r2 ** 2 * math.acos(ratio2) - 0.5 * math.sqrt(abs(a * b * c * e))
    return area / (math.pi * (min(r1, r2) ** 2))


def _blob_overlap(blob1, blob2):  # _blob_overlap for two image dimensions and one sigma column
    root = math.sqrt(2)
    if blob1[-1] == blob2[-1] == 0:
        return 0.0
    if blob1[-1] > blob2[-1]:
        max_sigma, r1, r2 = blob1[-1], 1.0, blob2[-1] / blob1[-1]
    else:
        max_sigma, r2, r1 = blob2[-1], 1.0, blob1[-1] / blob2[-1]
    pos1, pos2 = blob1[:2] / (max_sigma * root), blob2[:2] / (max_sigma * root)
    d = np.sqrt(np.sum((pos2 - pos1) ** 2))
    if d > r1 + r2:
        return 0.0
    if d <= abs(r1 - r2):
        return 1.0
    return _disk_overlap(d, r1, r2)


def _prune_blobs(blobs, overlap):
    """skimage/feature/blob.py ``_prune_blobs``: of two blobs whose discs (radius sigma * sqrt(2)) overlap by more than
    `overlap` of the smaller one the blob with the smaller sigma is dropped; the pairs come from the same
    ``cKDTree.query_pairs`` call, in the order a Python set yields them."""
    sigma = blobs[:, -1].max()
    distance = 2 * sigma * math.sqrt(2)
    tree = spatial.cKDTree(blobs[:, :-1])
    pairs = np.array(list(tree.query_pairs(distance)))
    if len(pairs) == 0:
        return blobs
    for (i, j) in pairs:
        blob1, blob2 = blobs[i], blobs[j]
        if _blob_overlap(blob1, blob2) > overlap:
            if blob1[-1] > blob2[-1]:
                blob2[-1] = 0
            else:
                blob1[-1] = 0
    return np.stack([b for b in blobs if b[-1] > 0])


def detection(input_image, max_num_features=None, method="log", threshold=0.5, min_sigma=3, max_sigma=20, overlap=0.5,
              return_sigmas=False, **kwargs):
    """Parameters and return value as documented for the reference (blob.py:44-96): ``(p, 2)`` pixel coordinates
    (x, y) of the detected blobs, with ``return_sigmas`` a third column holding the blobs' standard deviations."""
    if method not in ["log", "dog", "doh"]:
        raise ValueError("unknown method %s, must be 'log', 'dog' or 'doh'" % method)
    unsupported = None
    if method == "doh":
        unsupported = "method='doh'"
    elif not (np.isscalar(min_sigma) and np.isscalar(max_sigma)):
        unsupported = "anisotropic sigmas"
    elif set(kwargs) - {"num_sigma", "log_scale", "sigma_ratio"} or kwargs.get("exclude_border", False):
        unsupported = "keyword arguments %r" % sorted(kwargs)
    elif not isinstance(input_image, DeviceArray) and np.asarray(input_image).dtype not in (np.float32, np.float64):
        unsupported = "images of dtype %s" % np.asarray(input_image).dtype
    if unsupported is not None:
        try:
            from pysteps.feature.blob import detection as ref  # noqa: PLC0415
        except Exception as exc:
            raise NotImplementedError("pysteps_amd blob.detection: %s is not implemented on the HIP path" % unsupported) from exc
        if ref is detection or isinstance(input_image, DeviceArray):
            raise NotImplementedError("pysteps_amd blob.detection: %s is not implemented on the HIP path" % unsupported)
        warnings.warn("pysteps_amd blob.detection: %s -> delegating to the reference CPU path" % unsupported)
        return ref(input_image, max_num_features, method, threshold, min_sigma, max_sigma, overlap, return_sigmas, **kwargs)
    if len(input_image.shape) != 2:
        raise ValueError("input_image must be a two-dimensional array")

    image = _to_device(input_image)
    sigmas = _sigma_list(method, min_sigma, max_sigma, kwargs)
    cube = response_cube(image, sigmas, method)
    lm, _ = _peaks(cube, threshold)
    cube.free()
    if lm.shape[0] == 0:
        blobs = np.empty((0, 3))
    else:
        blobs = _prune_blobs(np.hstack([lm[:, :2].astype(np.float64), sigmas[lm[:, 2]][:, None]]), overlap)
    if max_num_features is not None and blobs.shape[0] > max_num_features:
        # blob.py:126-134: -gaussian_laplace(input_image, sigma) * sigma**2 at the blob's pixel - for "log" the cube's
        # own value there, for "dog" one Laplacian plane per distinct sigma among the blobs
        inten = np.empty(blobs.shape[0])
        for s in np.unique(blobs[:, 2]):
            sel = np.nonzero(blobs[:, 2] == s)[0]
            plane = response_cube(image, np.array([s]), "log")
            yx = np.ascontiguousarray(blobs[sel, :2].astype(np.int32))
            vals = np.empty(len(sel), dtype=np.float64)
            m, n = image.shape
            _lib.check(_lib.lib().psh_blob_gather_dev(plane.ptr, m, n, yx.ctypes.data, len(sel), vals.ctypes.data), "psh_blob_gather_dev")
            plane.free()
            inten[sel] = vals
        idx = np.argsort(inten, kind="stable")[::-1]
        blobs = blobs[idx[:max_num_features], :]
    if not return_sigmas:
        return np.column_stack([blobs[:, 1], blobs[:, 0]])
    return np.column_stack([blobs[:, 1], blobs[:, 0], blobs[:, 2]])
