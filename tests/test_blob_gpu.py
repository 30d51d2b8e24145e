"""The HIP blob detector (csrc/blob.hip, pysteps_amd/feature/blob.py) against the golden vectors of the unmodified
reference (pysteps/feature/blob.py:32-140 + scikit-image 0.18.3, tools/make_golden_blob.py), against SciPy itself for
the scale cube (bit for bit: SciPy's correlate1d arithmetic is reproduced operation by operation) and against the
oracle (oracle/blob.py) for the peak list, NaN regions included; then as ``fd_method="blob"`` of dense Lucas-Kanade
against the restated pipeline with the oracle's detector."""
import warnings

import numpy as np
import pytest
from scipy import ndimage as ndi

from helpers.blob_cases import load, same_blobs
from oracle import blob as oblob

pytestmark = pytest.mark.gpu

CASES, VERSIONS = load()


def _field(m, n, seed, nan=False, dtype=np.float64):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:m, 0:n].astype(np.float64)
    f = np.full((m, n), -15.0)
    for _ in range(max(6, m * n // 2500)):
        cy, cx, s, a = rng.uniform(0, m), rng.uniform(0, n), rng.uniform(2.0, 14.0), rng.uniform(10.0, 50.0)
        f = np.maximum(f, -15.0 + a * np.exp(-((y - cy) ** 2 + (x - cx) ** 2) / (2.0 * s * s)))
    f += rng.normal(0.0, 0.25, (m, n)) * (f > -14.5)
    if nan:
        f[(y + 0.5 * x) < 0.3 * m] = np.nan
        f[((y - 0.7 * m) ** 2 + (x - 0.6 * n) ** 2) < (0.07 * m) ** 2] = np.nan
        f[m // 3, n // 2] = np.nan  # an isolated missing pixel
    return f.astype(dtype)


@pytest.mark.parametrize("name,image,kw,want", CASES, ids=[c[0] for c in CASES])
def test_detection_reproduces_the_reference(name, image, kw, want):
    from pysteps_amd.feature.blob import detection

    got = detection(image, **kw)
    assert same_blobs(got, want), (name, VERSIONS, got, want)


@pytest.mark.parametrize("shape,nan,dtype", [((96, 130), False, np.float64), ((150, 90), True, np.float64), ((64, 64), True, np.float32),
                                             ((20, 300), False, np.float64), ((257, 255), True, np.float32)])
@pytest.mark.parametrize("method", ["log", "dog"])
def test_scale_cube_is_scipys_bit_for_bit(shape, nan, dtype, method):
    """cube[k] = -gaussian_laplace(image, s) * s**2 (LoG) / (G(s_k) - G(s_k+1)) * s_k (DoG) as SciPy computes them - also
    where the kernel is longer than the line (20 rows against 8 sigma + 1 = 161 taps: repeated reflection)."""
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.feature.blob import response_cube

    image = _field(shape[0], shape[1], 11, nan, dtype)
    sigmas = oblob.sigma_list_log(3, 20, 5) if method == "log" else oblob.sigma_list_dog(3, 20)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oblob.cube_log(image, sigmas) if method == "log" else oblob.cube_dog(image, sigmas)
    got = response_cube(DeviceArray.from_host(image, dtype=dtype), sigmas, method).to_host()
    want = np.moveaxis(np.asarray(want, dtype=np.float64), -1, 0)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    # NumPy 2 multiplies a float32 plane by the float64 scalar s**2 in float64, NumPy 1 in float32: the kernel follows NumPy 2
    if dtype == np.float64 or int(np.__version__.split(".")[0]) >= 2:
        assert np.array_equal(got[ok], want[ok]), np.abs(got[ok] - want[ok]).max()
    else:
        assert np.allclose(got[ok], want[ok], rtol=2e-7, atol=0.0)


@pytest.mark.parametrize("nan", [False, True])
def test_peaks_are_the_oracles_with_scipys_nan_semantics(nan):
    """maximum_filter's ring buffer decides what happens beside NaN: the peak list (coordinates and order) must be the one
    scipy.ndimage.maximum_filter + the mask of peak_local_max give on the same cube."""
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.feature.blob import _peaks

    image = _field(180, 200, 3, nan)
    sigmas = oblob.sigma_list_log(2, 12, 7)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cube = oblob.cube_log(image, sigmas)
        want = oblob.peaks(cube, 0.2)
    dev = DeviceArray.from_host(np.ascontiguousarray(np.moveaxis(cube, -1, 0)))
    got, values = _peaks(dev, 0.2)
    assert want.shape[0] > 5
    assert np.array_equal(got, want)
    assert np.array_equal(values, cube[tuple(want.T)])


def test_maximum_filter_next_to_nan_lines():
    """Synthetic cube with NaNs sprinkled in: every element of the three-pass maximum filter equals SciPy's, NaN for NaN."""
    from pysteps_amd import _lib
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.feature.blob import _peaks

    rng = np.random.default_rng(0)
    cube = rng.normal(0.0, 1.0, (40, 37, 6))
    cube[rng.random(cube.shape) < 0.08] = np.nan
    cube[5:9, 10:14, :] = 0.75  # plateaus: equal neighbours are all "maxima"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oblob.peaks(cube.copy(), -0.5)
    got, _ = _peaks(DeviceArray.from_host(np.ascontiguousarray(np.moveaxis(cube, -1, 0))), -0.5)
    # (ties in value: the oracle's argsort is not stable - compare as sets and the values' order)
    assert {tuple(r) for r in got} == {tuple(r) for r in want}
    assert np.array_equal(cube[tuple(got.T)], np.sort(cube[tuple(want.T)])[::-1])
    assert _lib.lib() is not None


def test_large_field_with_missing_regions_against_the_oracle():
    from pysteps_amd.feature.blob import detection

    image = _field(700, 900, 21, nan=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = oblob.detection(image, return_sigmas=True)
        want5 = oblob.detection(image, max_num_features=5, return_sigmas=True, method="dog")
    got = detection(image, return_sigmas=True)
    assert want.shape[0] > 20 and same_blobs(got, want)
    assert same_blobs(detection(image, max_num_features=5, return_sigmas=True, method="dog"), want5)


def test_device_resident_image_and_argument_checks():
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.feature.blob import detection

    image = _field(128, 160, 5).astype(np.float32)
    assert np.array_equal(detection(DeviceArray.from_host(image), return_sigmas=True), detection(image, return_sigmas=True))
    with pytest.raises(ValueError):
        detection(image, method="nosuch")
    with pytest.raises(NotImplementedError):
        detection(DeviceArray.from_host(image), method="doh")


def test_dense_lucaskanade_with_the_blob_detector():
    """fd_method="blob" (pysteps/motion/lucaskanade.py:191,230) end to end against the restated pipeline with the oracle's
    detector on the cleaned frame: same sparse vectors, same dense field."""
    from scipy.ndimage import gaussian_filter

    from oracle import lk_opencv as olk
    from pysteps_amd.motion import get_method

    rng = np.random.default_rng(2)
    base = _field(300, 340, 9)
    tex = gaussian_filter(rng.standard_normal(base.shape), 2.0) * 3.0 * (base > -14.0)
    f0 = base + tex
    f0[:40, :60] = np.nan
    frames = np.stack([f0, np.roll(f0, (2, 3), axis=(0, 1))])
    fd_kwargs = {"threshold": 0.3, "min_sigma": 2, "max_sigma": 10, "max_num_features": 60}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        det = lambda img: oblob.detection(img, **fd_kwargs)  # noqa: E731
        want_xy, want_uv = olk.dense_lucaskanade(frames, dense=False, detector=det)
        want = olk.dense_lucaskanade(frames, detector=det)
    xy, uv = get_method("LK")(frames, fd_method="blob", fd_kwargs=fd_kwargs, dense=False)
    assert xy.shape[0] > 10 and np.array_equal(xy, want_xy)
    assert np.abs(uv - want_uv).max() < 1e-2
    got = get_method("LK")(frames, fd_method="blob", fd_kwargs=fd_kwargs)
    assert got.shape == (2, 300, 340)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-3
    assert abs(np.median(uv[:, 0]) - 3.0) < 0.2 and abs(np.median(uv[:, 1]) - 2.0) < 0.2
