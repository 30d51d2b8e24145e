"""CPU restatement of the reference's blob feature detector (TEST INFRASTRUCTURE - never imported by the product).

``pysteps/feature/blob.py:32-140`` ``detection`` hands the image to scikit-image's ``blob_log`` / ``blob_dog``
(third party, scikit-image 0.18.3 in this project's images: ``skimage/feature/blob.py``, ``skimage/feature/peak.py``,
``skimage/_shared/coord.py``) and, with ``max_num_features``, keeps the blobs with the largest scale-normalised
Laplacian.  scikit-image is not importable under the interpreter the tests run with, so its algorithm is restated
here on top of the SciPy functions it calls itself (``scipy.ndimage.gaussian_laplace`` / ``gaussian_filter`` /
``maximum_filter``, ``scipy.spatial.cKDTree``):

  blob_log   sigma_list = linspace(min_sigma, max_sigma, num_sigma) (or logspace with log_scale);
             cube[..., k] = -gaussian_laplace(image, s_k) * s_k**2                              (blob.py:blob_log)
  blob_dog   sigma_k = min_sigma * ratio**k, k = 0 .. K, K = int(log(max/min) / log(ratio) + 1);
             cube[..., k] = (G(s_k) - G(s_k+1)) * s_k                                           (blob.py:blob_dog)
  peaks      peak_local_max(cube, threshold_abs=threshold, footprint=ones(3,3,3), threshold_rel=0,
             exclude_border=False): cube == maximum_filter(cube, 3x3x3, mode="constant") and cube > threshold,
             no peak at all if every element is its own neighbourhood maximum; coordinates in C order sorted by
             decreasing value; the min_distance=1 spacing step never removes distinct integer coordinates  (peak.py)
  pruning    pairs of blobs closer than 2 * max(sigma) * sqrt(2) (cKDTree.query_pairs); if the discs of radius
             sigma * sqrt(2) overlap by more than `overlap` of the smaller disc's area the blob with the smaller sigma
             gets sigma 0; blobs with sigma 0 are dropped at the end                           (blob.py:_prune_blobs)

Pinned by tests/golden/blob_reference.npz, written by the unmodified reference under /opt/conda/bin/python3.9
(tools/make_golden_blob.py).  Two things are left open by scikit-image itself and therefore here: the order of
exactly tied peak values (argsort) and, for chains of overlapping blobs, the order in which the pairs are visited
(iteration order of a Python set).
"""
import math

import numpy as np
from scipy import ndimage as ndi
from scipy import spatial


def sigma_list_log(min_sigma, max_sigma, num_sigma=10, log_scale=False):
    if log_scale:
        return np.logspace(np.log10(min_sigma), np.log10(max_sigma), num_sigma)
    return np.linspace(0, 1, num_sigma) * (float(max_sigma) - float(min_sigma)) + float(min_sigma)


def sigma_list_dog(min_sigma, max_sigma, sigma_ratio=1.6):
    k = int(np.mean(np.log(float(max_sigma) / float(min_sigma)) / np.log(sigma_ratio) + 1))
    return np.array([float(min_sigma) * (sigma_ratio ** i) for i in range(k + 1)])


def cube_log(image, sigmas):
    return np.stack([-ndi.gaussian_laplace(image, [s, s]) * np.float64(s) ** 2 for s in sigmas], axis=-1)


def cube_dog(image, sigmas):
    g = [ndi.gaussian_filter(image, [s, s]) for s in sigmas]
    return np.stack([(g[i] - g[i + 1]) * np.float64(sigmas[i]) for i in range(len(sigmas) - 1)], axis=-1)


def peaks(cube, threshold):
    """peak_local_max as blob_log / blob_dog call it -> (p, 3) int coordinates (row, col, scale index), strongest first."""
    image_max = ndi.maximum_filter(cube, footprint=np.ones((3, 3, 3)), mode="constant")
    out = cube == image_max
    if np.all(out):
        out[:] = False
    out &= cube > threshold
    coord = np.nonzero(out)
    idx = np.argsort(-cube[coord])
    return np.transpose(coord)[idx]


def _disk_overlap(d, r1, r2):
    ratio1 = min(max((d ** 2 + r1 ** 2 - r2 ** 2) / (2 * d * r1), -1), 1)
    ratio2 = min(max((d ** 2 + r2 ** 2 - r1 ** 2) / (2 * d * r2), -1), 1)
    a, b, c, e = -d + r2 + r1, d - r2 + r1, d + r2 - r1, d + r2 + r1
    area = r1 ** 2 * math.acos(ratio1) + r2 ** 2 * math.acos(ratio2) - 0.5 * math.sqrt(abs(a * b * c * e))
    return area / (math.pi * (min(r1, r2) ** 2))


def blob_overlap(blob1, blob2):
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


def prune(blobs, overlap):
    sigma = blobs[:, -1].max()
    distance = 2 * sigma * math.sqrt(2)
    tree = spatial.cKDTree(blobs[:, :-1])
    pairs = np.array(list(tree.query_pairs(distance)))
    if len(pairs) == 0:
        return blobs
    for (i, j) in pairs:
        blob1, blob2 = blobs[i], blobs[j]
        if blob_overlap(blob1, blob2) > overlap:
            if blob1[-1] > blob2[-1]:
                blob2[-1] = 0
            else:
                blob1[-1] = 0
    return np.stack([b for b in blobs if b[-1] > 0]) if np.any(blobs[:, -1] > 0) else np.empty((0, 3))


def detection(input_image, max_num_features=None, method="log", threshold=0.5, min_sigma=3, max_sigma=20, overlap=0.5,
              return_sigmas=False, **kwargs):
    """pysteps/feature/blob.py:32-140 for method 'log' / 'dog'."""
    if method not in ["log", "dog", "doh"]:
        raise ValueError("unknown method %s, must be 'log', 'dog' or 'doh'" % method)
    if method == "doh":
        raise NotImplementedError("oracle: determinant-of-Hessian blobs are not restated")
    image = np.asarray(input_image)
    if image.dtype not in (np.float32, np.float64):
        image = image.astype(np.float64)
    if method == "log":
        sigmas = sigma_list_log(min_sigma, max_sigma, kwargs.get("num_sigma", 10), kwargs.get("log_scale", False))
        cube = cube_log(image, sigmas)
    else:
        sigmas = sigma_list_dog(min_sigma, max_sigma, kwargs.get("sigma_ratio", 1.6))
        cube = cube_dog(image, sigmas)
    lm = peaks(cube, threshold)
    if lm.size == 0:
        blobs = np.empty((0, 3))
    else:
        blobs = prune(np.hstack([lm[:, :2].astype(np.float64), np.asarray(sigmas)[lm[:, 2]][:, None]]), overlap)
    if max_num_features is not None and blobs.shape[0] > max_num_features:
        # blob.py:126-134: the scale-normalised Laplacian at the blob's own sigma, whatever the method was
        inten = []
        for i in range(blobs.shape[0]):
            gl = -ndi.gaussian_laplace(input_image, blobs[i, 2]) * blobs[i, 2] ** 2
            inten.append(gl[int(blobs[i, 0]), int(blobs[i, 1])])
        idx = np.argsort(inten)[::-1]
        blobs = blobs[idx[:max_num_features], :]
    if not return_sigmas:
        return np.column_stack([blobs[:, 1], blobs[:, 0]])
    return np.column_stack([blobs[:, 1], blobs[:, 0], blobs[:, 2]])
