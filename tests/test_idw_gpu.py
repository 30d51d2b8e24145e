"""HIP k-NN IDW interpolation vs the reference's goldens and the cKDTree oracle.

Mirrors pysteps/tests/test_utils_interpolate.py (shapes, finiteness, error cases,
single sample / uniform values -> uniform field, k=1, k=None).  Tolerance: the
kernel computes in float32 -> relative L2 <= 1e-5 against the float64 reference,
max abs <= 1e-4 px/step; pixels whose k-th and (k+1)-th neighbours are exactly
equidistant are excluded (the tie order is implementation-defined in cKDTree too).
"""

import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def idw():
    from pysteps_amd.utils import idwinterp2d

    return idwinterp2d


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "sparse_reference.npz"))


def _tie_mask(xy, m, n, k):
    """Pixels whose k-th and (k+1)-th nearest samples are (numerically) equidistant: there the
    selected set is implementation-defined (cKDTree's order is arbitrary as well)."""
    from scipy.spatial import cKDTree

    if k is None or k >= len(xy):
        return np.zeros((m, n), dtype=bool)
    gx, gy = np.meshgrid(np.arange(n), np.arange(m))
    d, _ = cKDTree(xy).query(np.column_stack([gx.ravel(), gy.ravel()]), k=k + 1)
    # float32 sample coordinates move distances by up to ~1e-4 px at |x| ~ 1e3
    return (d[:, k] - d[:, k - 1] <= 2e-4).reshape(m, n)


def _close(got, want, ties=None):
    assert got.shape == want.shape and got.dtype == np.float64
    assert np.isfinite(got).all()
    ok = np.ones(got.shape, dtype=bool) if ties is None else np.broadcast_to(~ties, got.shape)
    assert ok.mean() > 0.98
    assert np.max(np.abs(got - want)[ok]) <= 1e-4
    assert rel_l2(got[ok], want[ok]) < 1e-5


@pytest.mark.parametrize("case", ["a", "b"])
def test_matches_reference_golden(idw, gold, case):
    m, n = gold[case + "/shape"]
    dxy, duv = gold[case + "/dxy"], gold[case + "/duv"]
    _close(idw(dxy, duv, np.arange(n), np.arange(m)), gold[case + "/idw"], _tie_mask(dxy, m, n, 20))
    _close(idw(dxy, duv, np.arange(n), np.arange(m), power=2.0, k=5, dist_offset=0.1), gold[case + "/idw_k5_p2"],
           _tie_mask(dxy, m, n, 5))


@pytest.mark.parametrize("L,m,n,k", [(2, 17, 33, 20), (19, 40, 50, 20), (21, 40, 50, 20), (300, 333, 257, 20),
                                     (300, 100, 100, 1), (300, 100, 100, 32), (50, 64, 64, None), (2500, 512, 512, 20)])
def test_vs_oracle_random_positions(idw, L, m, n, k):
    from oracle import sparse as osp

    rng = np.random.default_rng(L + m)
    xy = np.column_stack([rng.uniform(-5, n + 5, L), rng.uniform(-5, m + 5, L)])  # no ties
    uv = rng.normal(0, 2, (L, 2))
    want = osp.idw(xy, uv, m, n, k=k)
    got = idw(xy, uv, np.arange(n), np.arange(m), k=k)
    _close(got, want, _tie_mask(xy, m, n, k))


def test_clustered_samples_overflow_path(idw):
    """More candidates than the LDS list holds -> exact brute-force path."""
    from oracle import sparse as osp

    rng = np.random.default_rng(4)
    L, m, n = 1800, 96, 96
    xy = np.column_stack([rng.normal(48, 3, L), rng.normal(48, 3, L)])
    uv = rng.normal(0, 1, (L, 2))
    _close(idw(xy, uv, np.arange(n), np.arange(m)), osp.idw(xy, uv, m, n), _tie_mask(xy, m, n, 20))


def test_fine_tile_overflow_and_variants_agree(idw):
    """Moderately clustered samples: the supertile lists fit, some 8x8 tiles overflow their LDS list
    (brute-force path per tile); and the one-level kernel (idw_variant 1) gives the same field."""
    from oracle import sparse as osp
    from pysteps_amd import _lib

    rng = np.random.default_rng(9)
    m, n = 200, 260
    xy = np.concatenate([np.column_stack([rng.normal(60, 6, 150), rng.normal(70, 6, 150)]),
                         np.column_stack([rng.uniform(0, n, 400), rng.uniform(0, m, 400)])])
    uv = rng.normal(0, 1, (xy.shape[0], 2))
    want = osp.idw(xy, uv, m, n)
    got = idw(xy, uv, np.arange(n), np.arange(m))
    _close(got, want, _tie_mask(xy, m, n, 20))
    lib = _lib.lib()
    _lib.check(lib.psh_set_option(b"idw_variant", 1))
    try:
        old = idw(xy, uv, np.arange(n), np.arange(m))
    finally:
        _lib.check(lib.psh_set_option(b"idw_variant", 0))
    _close(old, want, _tie_mask(xy, m, n, 20))
    keep = ~_tie_mask(xy, m, n, 20)
    assert np.max(np.abs(old - got)[:, keep]) < 1e-5


def test_trivial_cases_and_shapes(idw):
    xg, yg = np.arange(30), np.arange(20)
    one = idw(np.array([[3.0, 4.0]]), np.array([[1.5, -2.0]]), xg, yg)
    assert one.shape == (2, 20, 30) and np.all(one[0] == 1.5) and np.all(one[1] == -2.0)
    same = idw(np.array([[3.0, 4.0], [9.0, 1.0]]), np.array([2.0, 2.0]), xg, yg)
    # decorators.py:207-208 returns this case without squeezing
    assert same.shape == (1, 20, 30) and np.all(same == 2.0)
    rng = np.random.default_rng(0)
    xy = rng.uniform(0, 20, (10, 2))
    assert idw(xy, rng.normal(size=10), xg, yg).shape == (20, 30)
    assert idw(xy, rng.normal(size=(10, 1)), xg, yg).shape == (20, 30)
    assert idw(xy, rng.normal(size=(10, 3)), xg, yg).shape == (3, 20, 30)
    shifted = idw(xy, rng.normal(size=(10, 2)), np.arange(5, 35) * 2.0, np.arange(20) * 2.0)
    assert shifted.shape == (2, 20, 30) and np.isfinite(shifted).all()


def test_scaled_grid_vs_oracle(idw):
    """Non-unit grid spacing: distances are divided by the mean resolution (interpolate.py:89-93)."""
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(3)
    xy = rng.uniform(0, 100, (40, 2))
    uv = rng.normal(size=(40, 2))
    xg, yg = np.arange(0, 100, 2.0), np.arange(0, 60, 2.0)
    gx, gy = np.meshgrid(xg, yg)
    d, i = cKDTree(xy).query(np.column_stack([gx.ravel(), gy.ravel()]), k=20)
    w = 1.0 / np.sqrt(d / 2.0 + 0.5)
    w /= w.sum(axis=1, keepdims=True)
    want = np.moveaxis((uv[i] * w[..., None]).sum(axis=1).reshape(yg.size, xg.size, 2), -1, 0)
    _close(idw(xy, uv, xg, yg), want)


def test_errors(idw):
    xg, yg = np.arange(10), np.arange(10)
    xy = np.random.default_rng(0).uniform(0, 10, (6, 2))
    with pytest.raises(ValueError):
        idw(xy, np.array([1.0, np.nan, 1, 1, 1, 1]), xg, yg)
    with pytest.raises(ValueError):
        idw(xy, np.ones((5, 2)), xg, yg)
    with pytest.raises(ValueError):
        idw(np.ones(6), np.arange(6.0), xg, yg)
    with pytest.raises(ValueError):
        idw(xy, np.ones((6, 2, 2)), xg, yg)


# ---- device outlier test (csrc/sparse_qc.hip) ------------------------------------
@pytest.mark.parametrize("n,k", [(2, 30), (5, 30), (31, 30), (32, 30), (900, 30), (400, 5), (300, 60)])
def test_device_outliers_match_host(n, k):
    from pysteps_amd.utils import detect_outliers, detect_outliers_device

    rng = np.random.default_rng(n + k)
    xy = rng.uniform(0, 4096, (n, 2))  # continuous positions: no equidistant neighbours
    uv = rng.normal(0, 1, (n, 2)) + [3, -2]
    uv[rng.integers(0, n, max(1, n // 30))] += rng.uniform(4, 9, 2)
    want = detect_outliers(uv, 3, xy, k)
    got = detect_outliers_device(uv, 3, xy, k)
    assert got.dtype == bool and np.array_equal(got, want)


def test_device_outliers_reference_golden(gold):
    """integer feature positions (ties possible): at most a handful of borderline flips."""
    from pysteps_amd.utils import detect_outliers_device

    for case in ("a", "b", "c"):
        got = detect_outliers_device(gold[case + "/uv"], 3, gold[case + "/xy"], 30)
        want = gold[case + "/outliers"]
        assert np.count_nonzero(got != want) <= max(1, 0.01 * want.size)


def test_device_outliers_degenerate():
    from pysteps_amd.utils import detect_outliers_device

    assert not detect_outliers_device(np.zeros((20, 2)), 1, np.random.default_rng(0).uniform(0, 9, (20, 2)), 5).any()
    assert detect_outliers_device(np.zeros((1, 2)), 1, np.zeros((1, 2)), 5).shape == (1,)
    assert detect_outliers_device(np.zeros((0, 2)), 1, np.zeros((0, 2)), 5).shape == (0,)


def test_idw_4096_sampled_pixels_vs_oracle():
    """Full-size field (4096^2, ~1000 declustered vectors): the cKDTree oracle is evaluated on a
    random sample of pixels only (it needs 100 s for the whole grid)."""
    from scipy.spatial import cKDTree

    from pysteps_amd.utils.interpolate import idw_to_device

    m = n = 4096
    rng = np.random.default_rng(12)
    L = 1000
    xy = np.column_stack([rng.integers(0, n, L), rng.integers(0, m, L)]).astype(float) + 0.5 * rng.integers(0, 2, (L, 2))
    uv = np.column_stack([4 + 2 * np.sin(2 * np.pi * xy[:, 1] / m), -3 + 1.5 * np.cos(2 * np.pi * xy[:, 0] / n)])
    uv += rng.normal(0, 0.1, uv.shape)
    field = idw_to_device(xy, uv, m, n).to_host()
    assert field.shape == (2, m, n) and np.isfinite(field).all()
    sx, sy = rng.integers(0, n, 4000), rng.integers(0, m, 4000)
    # corners and edges as well
    sx[:4], sy[:4] = [0, n - 1, 0, n - 1], [0, 0, m - 1, m - 1]
    d, i = cKDTree(xy).query(np.column_stack([sx, sy]).astype(float), k=21)
    ties = d[:, 20] - d[:, 19] <= 2e-4
    d, i = d[:, :20], i[:, :20]
    w = 1.0 / np.sqrt(d + 0.5)
    w /= w.sum(axis=1, keepdims=True)
    want = (uv[i] * w[..., None]).sum(axis=1)
    got = field[:, sy, sx].T
    assert np.max(np.abs(got - want)[~ties]) < 1e-4
