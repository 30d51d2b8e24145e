"""Golden vectors for the blob feature detector, written by the UNMODIFIED reference.

    /opt/conda/bin/python3.9 tools/make_golden_blob.py        (-> tests/golden/blob_reference.npz)

``pysteps/feature/blob.py`` ``detection`` is a thin layer over scikit-image's ``blob_log`` / ``blob_dog``
(third party, absent from the interpreter the tests run with; this container holds scikit-image 0.18.3 with
SciPy 1.7.1 / NumPy 1.26.4 under /opt/conda).  The reference module is loaded from /root/reference by path
(tools/ref_loader.py); inputs are synthetic precipitation-like fields (dBR-like values, smooth cells of several
sizes, optionally a NaN wedge as outside a radar's range).  Every case stores its input, keyword arguments and
the array the reference returns (images as int16 counts of 1/64, NaN = -32768: tests/helpers unpack them).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import ref_loader  # noqa: E402


def field(m, n, seed, ncells=14, nan_wedge=False, dtype=np.float64):
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:m, 0:n].astype(np.float64)
    f = np.full((m, n), -15.0)
    for _ in range(ncells):
        cy, cx = rng.uniform(10, m - 10), rng.uniform(10, n - 10)
        s = rng.uniform(2.5, 16.0)
        a = rng.uniform(15.0, 50.0)
        f = np.maximum(f, -15.0 + a * np.exp(-((y - cy) ** 2 + (x - cx) ** 2) / (2.0 * s * s)))
    f += rng.normal(0.0, 0.3, size=(m, n)) * (f > -14.0)
    if nan_wedge:
        f[(y + 0.6 * x) < 0.35 * m] = np.nan
        f[((y - 0.8 * m) ** 2 + (x - 0.75 * n) ** 2) < (0.08 * m) ** 2] = np.nan
    # quantised to 1/64 (exact in float32 and float64): the file stores int16 counts, NaN as -32768
    f = np.round(f * 64.0) / 64.0
    return f.astype(dtype)


def pack(img):
    q = np.where(np.isnan(img), -32768, np.nan_to_num(img.astype(np.float64)) * 64.0)
    assert np.all(q == np.round(q)) and np.all(np.abs(q) <= 32768)
    return q.astype(np.int16)


def main():
    blob = ref_loader.load("pysteps.feature.blob")
    assert blob.SKIMAGE_IMPORTED
    import skimage, scipy  # noqa: E401

    cases = [
        ("default", field(200, 200, 1), {}),
        ("sigmas", field(200, 200, 1), {"return_sigmas": True}),
        ("rect_sigmas", field(150, 230, 2, ncells=18), {"return_sigmas": True}),
        ("nan_wedge", field(220, 200, 3, nan_wedge=True), {"return_sigmas": True}),
        ("max5", field(200, 200, 1), {"max_num_features": 5, "return_sigmas": True}),
        ("max3_nan", field(220, 200, 3, nan_wedge=True), {"max_num_features": 3}),
        ("threshold_low", field(160, 180, 4), {"threshold": 0.1, "min_sigma": 2, "max_sigma": 12, "return_sigmas": True}),
        ("overlap_tight", field(200, 200, 5, ncells=25), {"overlap": 0.1, "return_sigmas": True}),
        ("num_sigma_log_scale", field(180, 160, 6), {"num_sigma": 6, "log_scale": True, "return_sigmas": True}),
        ("nothing", field(120, 120, 7), {"threshold": 1e6, "return_sigmas": True}),
        ("uniform", np.full((90, 110), 3.25), {"return_sigmas": True}),
        ("float32", field(200, 200, 8, dtype=np.float32), {"return_sigmas": True}),
        ("dog", field(200, 200, 1), {"method": "dog", "return_sigmas": True}),
        ("dog_nan", field(220, 200, 3, nan_wedge=True), {"method": "dog", "threshold": 0.3, "return_sigmas": True}),
        ("dog_max4", field(200, 200, 9), {"method": "dog", "max_num_features": 4}),
    ]
    out = {"versions": json.dumps({"skimage": skimage.__version__, "scipy": scipy.__version__, "numpy": np.__version__}),
           "names": np.array([c[0] for c in cases])}
    for name, img, kw in cases:
        res = blob.detection(img.copy(), **kw)
        out[name + "__image_q64"] = pack(img)
        out[name + "__dtype"] = str(img.dtype)
        out[name + "__kwargs"] = json.dumps(kw)
        out[name + "__points"] = np.asarray(res, dtype=np.float64)
        print("%-22s %s -> %s" % (name, img.shape, np.asarray(res).shape))
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "blob_reference.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
