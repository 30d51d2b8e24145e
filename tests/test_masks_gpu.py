"""Device incremental mask (csrc/mask.hip, psh_dilated_mask_dev) against the reference's
compute_dilated_mask (pysteps/nowcasts/utils.py:69-101, from oracle/_ref) and the oracle: small
integers and one exact division, so the bar is bit-exact."""

import warnings

import numpy as np
import pytest

from oracle import masks as oracle

pytestmark = pytest.mark.gpu


def _structures():
    from scipy.ndimage import generate_binary_structure, iterate_structure

    cross = generate_binary_structure(2, 1)
    return [cross, iterate_structure(cross, 2), iterate_structure(cross, 3), np.ones((3, 3), bool),
            np.array([[1, 0, 0], [0, 1, 1], [0, 0, 0]], bool), np.ones((2, 4), bool),
            np.array([[0, 1, 1, 0, 1]], bool)]


def _rain_mask(shape, seed, fraction):
    from scipy.ndimage import gaussian_filter

    g = gaussian_filter(np.random.default_rng(seed).standard_normal(shape), 5.0)
    return g > np.quantile(g, 1.0 - fraction)


@pytest.mark.parametrize("shape", [(64, 64), (70, 130), (257, 255), (1024, 1024)])
def test_bit_exact_with_the_reference(ref_pysteps, shape):
    from pysteps.nowcasts.utils import compute_dilated_mask as ref

    from pysteps_amd.nowcasts.utils import compute_dilated_mask

    structures = _structures()
    for it, (fraction, r) in enumerate([(0.2, 10), (0.02, 3), (0.5, 0), (0.001, 25), (0.2, 1), (0.05, 10), (0.3, 4)]):
        mask = _rain_mask(shape, it + shape[0], fraction)
        kr = structures[it % len(structures)]
        want = ref(mask, kr, r)
        got = compute_dilated_mask(mask, kr, r)
        assert got.dtype == want.dtype and got.shape == want.shape
        assert np.array_equal(got, want), (it, r)
        assert np.array_equal(got, oracle.compute_dilated_mask(mask, kr, r))


def test_resident_small_empty_and_numeric_masks(ref_pysteps):
    from pysteps.nowcasts.utils import compute_dilated_mask as ref

    from pysteps_amd.device import DeviceArray
    from pysteps_amd.nowcasts.utils import compute_dilated_mask

    rng = np.random.default_rng(3)
    cross = _structures()[0]
    for shape in [(1, 1), (1, 9), (7, 1), (5, 6), (33, 65)]:  # resident masks always take the kernels
        mask = rng.random(shape) < 0.2
        got = compute_dilated_mask(DeviceArray.from_host(mask.astype(np.uint8)), cross, 3)
        assert isinstance(got, DeviceArray)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert np.array_equal(got.to_host(), ref(mask, cross, 3), equal_nan=True), shape
    empty = np.zeros((80, 90), bool)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # 0 / 0
        assert np.isnan(compute_dilated_mask(empty, cross, 10)).all() and np.isnan(ref(empty, cross, 10)).all()
    full = np.ones((80, 90), bool)
    assert np.array_equal(compute_dilated_mask(full, cross, 10), ref(full, cross, 10))
    values = rng.random((80, 90)) * 3  # cast to uint8 first (:87): 0.5 -> 0
    assert np.array_equal(compute_dilated_mask(values, cross, 5), ref(values, cross, 5))
    big = np.ones((41, 41), bool)  # 1681 set elements: the reference's function answers
    mask = rng.random((80, 90)) < 0.01
    assert np.array_equal(compute_dilated_mask(mask, big, 2), ref(mask, big, 2))
    with pytest.raises(NotImplementedError):
        compute_dilated_mask(DeviceArray.from_host(mask.astype(np.uint8)), big, 2)


@pytest.mark.parametrize("shape", [(64, 64), (70, 130), (257, 255), (1024, 1024), (40, 41), (1, 77)])
def test_bit_mask_entry_point_is_bit_exact_with_the_reference(ref_pysteps, shape):
    """psh_steps_incremental_mask_dev - `field >= thr` and the dilated mask in two kernels on bit masks (what the
    resident STEPS update calls per member) - against the reference's compute_dilated_mask of the thresholded field:
    structures with their centre element, rims up to the tile halo; NaNs in the field; empty and full masks; and the
    structures / rims the entry point declines (PSH_EUNSUPPORTED: the byte-mask kernels answer those)."""
    import ctypes

    from pysteps.nowcasts.utils import compute_dilated_mask as ref

    from pysteps_amd import _lib
    from pysteps_amd.device import DeviceArray

    lib = _lib.lib()
    m, n = shape
    structures = _structures()
    rng = np.random.default_rng(m * 1000 + n)

    def run(field, kr, r, thr):
        kr8 = np.ascontiguousarray(np.asarray(kr) != 0, dtype=np.uint8)
        out = DeviceArray((m, n), np.float64)
        rc = lib.psh_steps_incremental_mask_dev(DeviceArray.from_host(field).ptr, m, n, float(thr),
                                                kr8.ctypes.data_as(ctypes.c_void_p), kr8.shape[0], kr8.shape[1], int(r), out.ptr)
        return rc, out

    from scipy.ndimage import gaussian_filter

    for it, (fraction, r) in enumerate([(0.2, 10), (0.02, 3), (0.5, 0), (0.001, 20), (0.2, 1), (0.05, 14), (0.3, 4)]):
        g = gaussian_filter(rng.standard_normal(shape), 3.0) if min(shape) > 8 else rng.standard_normal(shape)
        thr = np.quantile(g, 1.0 - fraction)
        field = g.astype(np.float64)
        field[rng.random(shape) < 0.01] = np.nan  # NaN >= thr is False
        kr = structures[it % len(structures)]
        rc, out = run(field, kr, r, thr)
        has_centre = bool(np.asarray(kr)[kr.shape[0] // 2, kr.shape[1] // 2])
        reach = max(kr.shape[0] // 2, kr.shape[1] // 2, (kr.shape[0] - 1) - kr.shape[0] // 2, (kr.shape[1] - 1) - kr.shape[1] // 2)
        if not has_centre:
            assert rc == _lib.PSH_EUNSUPPORTED
            continue
        _lib.check(rc, "psh_steps_incremental_mask_dev")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = ref(field >= thr, kr, r)
        assert np.array_equal(out.to_host(), want, equal_nan=True), (it, r, reach)
    cross = structures[0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # 0 / 0
        rc, out = run(np.zeros(shape), cross, 10, 1.0)
        _lib.check(rc, "empty")
        assert np.isnan(out.to_host()).all()
    rc, out = run(np.ones(shape), cross, 10, 1.0)
    _lib.check(rc, "full")
    assert np.array_equal(out.to_host(), ref(np.ones(shape, bool), cross, 10))
    rc, _ = run(np.ones(shape), cross, 24, 1.0)  # rim + reach = 25 > 24
    assert rc == _lib.PSH_EUNSUPPORTED
