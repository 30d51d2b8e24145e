"""Behaviour at the edges of the device paths (round-2 advisor findings): inconsistent filter
dictionaries, stale weight caches, float64 inputs beyond the float32 range, thresholds that are not
float32 numbers, resident inputs the bucket pass of the CDF matching declines."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bp(levels, m, n, seed=0):
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.1, 1.0, (levels, m, n // 2 + 1))
    return {"weights_2d": w / w.sum(axis=0, keepdims=True), "weights_1d": np.zeros((levels, 4)), "shape": (m, n)}


def test_inconsistent_filter_dictionaries_raise_like_numpy_would():
    from pysteps_amd.cascade import decomposition_fft
    from pysteps_amd.noise.fftgenerators import generate_noise_2d_fft_filter

    field = np.random.default_rng(1).standard_normal((64, 64))
    bp = _bp(4, 64, 64)
    bp["weights_1d"] = np.zeros((5, 4))  # one level more than weights_2d holds
    with pytest.raises(ValueError):
        decomposition_fft(field, bp, normalize=True)
    F = {"field": np.ones((64, 32)), "input_shape": (64, 64), "use_full_fft": False}  # 33 columns expected
    with pytest.raises(ValueError):
        generate_noise_2d_fft_filter(F, randstate=np.random.RandomState(1))


def test_weight_cache_notices_in_place_edits_and_resident_nans():
    from pysteps_amd.cascade import decomposition_fft
    from pysteps_amd.cascade import decomposition as mod
    from pysteps_amd.device import DeviceArray

    field = np.random.default_rng(2).standard_normal((64, 128))
    bp = _bp(3, 64, 128)
    a = decomposition_fft(field, bp)["cascade_levels"]
    bp["weights_2d"][0] *= 0.5  # same array object, new content
    b = decomposition_fft(field, bp)["cascade_levels"]
    np.testing.assert_allclose(b[0], 0.5 * a[0], rtol=1e-12, atol=1e-14)
    mod.invalidate_weights_cache()
    assert not mod._weights_cache
    bad = field.copy()
    bad[5, 7] = np.nan
    with pytest.raises(ValueError):  # decomposition.py:195-196, checked on the device for resident fields
        decomposition_fft(DeviceArray.from_host(bad), bp)


def test_float64_inputs_are_checked_before_they_are_narrowed():
    """a finite float64 value beyond the float32 range is not a non-finite input (semilagrangian.py:106-137
    looks at the array it was given)"""
    from pysteps_amd import extrapolation

    ex = extrapolation.get_method("semilagrangian")
    P = np.random.default_rng(3).uniform(0, 10, (96, 128))
    P[40, 60] = 1e39  # finite in float64, infinite after narrowing
    V = np.zeros((2, 96, 128))
    out = ex(P, V, 1)
    assert out.shape == (1, 96, 128)
    far = np.ones_like(P, dtype=bool)
    far[38:43, 58:63] = False
    np.testing.assert_allclose(out[0][far], P[far], rtol=1e-6)
    P[10, 10] = np.inf  # a real non-finite value still is one
    with pytest.raises(ValueError):
        ex(P, V, 1)


def test_check_norain_threshold_rounding_follows_numpy():
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.utils import check_norain

    thr = 0.1  # not a float32 number: float32(0.1) > 0.1
    arr = np.full((64, 64), np.float32(0.1), dtype=np.float32)
    for t in (thr, np.float32(thr), np.float64(thr)):
        want = np.count_nonzero(arr > t) / arr.size <= 0.5
        assert check_norain(DeviceArray.from_host(arr), t, 0.5, None, False) == want
    assert np.count_nonzero(arr > np.float64(thr)) == arr.size and np.count_nonzero(arr > thr) == 0  # the two NumPy answers differ


def test_resident_probmatch_falls_back_when_the_bucket_pass_declines(ref_pysteps):
    from pysteps.postprocessing.probmatching import nonparam_match_empirical_cdf as ref

    from pysteps_amd.device import DeviceArray
    from pysteps_amd.postprocessing.probmatching import nonparam_match_empirical_cdf

    rng = np.random.default_rng(5)
    init = rng.standard_normal((256, 256))
    init[:100] = np.round(init[:100], 1)  # 25600 values on ~60 distinct numbers: tie groups > 16384? no - crowd one bucket:
    init[100:200] = 0.123456  # 25600 tied wet values
    target = np.where(rng.uniform(size=init.shape) < 0.4, rng.gamma(2.0, 2.0, init.shape), 0.0)
    got = nonparam_match_empirical_cdf(DeviceArray.from_host(init), DeviceArray.from_host(target))
    want = ref(init, target)
    assert isinstance(got, DeviceArray)
    g = got.to_host()
    # tied values may receive their target values in any order (the reference's argsort is unstable)
    untied = np.ones(init.shape, bool)
    untied[100:200] = False
    untied[:100] = False
    np.testing.assert_array_equal(g[untied], want[untied])
    np.testing.assert_array_equal(np.sort(g[~untied]), np.sort(want[~untied]))


def test_cached_step_factors_survive_a_wrapped_constant_ring():
    """Round-4 advisor: the extrapolator keeps the device copy of its step factors while they do not change.
    More than 64 calls of other entry points that take constant slots (here: dilated masks) between two
    extrapolations with equal time steps must not change what the second one advects with."""
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.extrapolation import get_method
    from pysteps_amd.nowcasts.utils import compute_dilated_mask
    from tools import synth

    m, n = 128, 160
    p = synth.rain_field_db(m, n, seed=5)
    v = synth.true_velocity(m, n)
    steps = [0.5, 1.0, 2.5, 3.0]
    ex = get_method("semilagrangian")
    first = ex(p, v, steps)
    mask = DeviceArray.from_host((p > -10).astype(np.uint8))
    kr = np.ones((3, 3), dtype=np.uint8)
    for _ in range(70):  # the ring has 64 slots: the one the factors used to share is overwritten by taps
        compute_dilated_mask(mask, kr, 4)
    second = ex(p, v, steps)
    assert np.array_equal(first, second, equal_nan=True)
