"""The blob detector's oracle (oracle/blob.py: scikit-image's blob_log / blob_dog restated on SciPy) against the
golden vectors the UNMODIFIED reference wrote (pysteps/feature/blob.py:32-140 with scikit-image 0.18.3 under
/opt/conda/bin/python3.9, tools/make_golden_blob.py), and the host pieces of the HIP detector that need no GPU."""
import warnings

import numpy as np
import pytest
from scipy import ndimage as ndi

from helpers.blob_cases import load, same_blobs
from oracle import blob as oblob

CASES, VERSIONS = load()


@pytest.mark.parametrize("name,image,kw,want", CASES, ids=[c[0] for c in CASES])
def test_oracle_reproduces_the_reference(name, image, kw, want):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = oblob.detection(image, **kw)
    assert same_blobs(got, want), (name, VERSIONS)


def test_golden_covers_what_the_reference_tests_cover():
    # pysteps/tests/test_feature.py: blob on a real composite with max_num_features None / 5; here: both, NaN regions,
    # both methods, float32, empty and trivial results
    names = {c[0] for c in CASES}
    assert {"default", "max5", "nan_wedge", "dog", "float32", "nothing", "uniform"} <= names
    assert any(np.isnan(c[1]).any() for c in CASES) and any(c[3].shape[0] == 0 for c in CASES)


@pytest.mark.parametrize("sigma", [1.0, 3.0, 4.888888888888889, 20.0])
def test_kernel_halves_are_scipys_weights(sigma):
    """The weights handed to psh_blob_cube_dev: a delta image filtered by scipy.ndimage.gaussian_filter1d returns them."""
    from pysteps_amd.feature.blob import _half_kernels

    radii, w = _half_kernels([sigma])
    r = int(radii[0])
    assert r == int(4.0 * sigma + 0.5) and w.shape == (2 * (r + 1),)
    delta = np.zeros(2 * r + 1)
    delta[r] = 1.0
    for order, half in ((0, w[: r + 1]), (2, w[r + 1:])):
        full = ndi.gaussian_filter1d(delta, sigma, order=order, mode="constant")
        assert np.array_equal(half, full[r::-1]) or np.allclose(half, full[r::-1], rtol=1e-15, atol=1e-300)


def test_sigma_lists():
    from pysteps_amd.feature.blob import _sigma_list

    assert np.array_equal(_sigma_list("log", 3, 20, {}), oblob.sigma_list_log(3, 20))
    assert np.array_equal(_sigma_list("log", 2, 12, {"num_sigma": 6, "log_scale": True}), oblob.sigma_list_log(2, 12, 6, True))
    assert np.array_equal(_sigma_list("dog", 3, 20, {}), oblob.sigma_list_dog(3, 20))


def test_prune_matches_the_oracle():
    from pysteps_amd.feature.blob import _prune_blobs

    rng = np.random.default_rng(5)
    blobs = np.column_stack([rng.integers(0, 120, 80), rng.integers(0, 120, 80), rng.choice([3.0, 4.9, 8.7, 12.4, 20.0], 80)]).astype(float)
    assert np.array_equal(_prune_blobs(blobs.copy(), 0.5), oblob.prune(blobs.copy(), 0.5))


def test_interface_names():
    from pysteps_amd.feature import get_method
    from pysteps_amd.feature import blob, shitomasi

    assert get_method("blob") is blob.detection and get_method("ShiTomasi") is shitomasi.detection
    with pytest.raises(ValueError):
        get_method("nosuch")
    with pytest.raises(ValueError):
        blob.detection(np.zeros((8, 8)), method="nosuch")
