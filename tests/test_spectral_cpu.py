"""The spectral mirrors on shapes the HIP kernels do not take (no GPU needed): they have to hand
the call to the reference's own code and return exactly its result - the drop-in contract for every
grid beyond the kernels (a side longer than 4096 that is not a power of two; since round 3 every other
shape, pysteps' 200 x 200 test fields included, runs on the device)."""

import numpy as np
import pytest


def _field(shape, seed):
    rng = np.random.default_rng(seed)
    from scipy.ndimage import gaussian_filter

    g = gaussian_filter(rng.standard_normal(shape), 4.0)
    return np.where(g > 0, 10.0 * g / g.std(), -15.0)


def test_decomposition_on_other_shapes_is_the_reference(ref_pysteps):
    from pysteps.cascade.bandpass_filters import filter_gaussian
    from pysteps.cascade.decomposition import decomposition_fft as ref_decomp
    from pysteps.cascade.decomposition import recompose_fft as ref_recomp

    from pysteps_amd.cascade import decomposition_fft, recompose_fft

    shape = (4100, 12)
    field = _field(shape, 1)
    bp = filter_gaussian(shape, 5)
    for kw in (dict(normalize=True, compute_stats=True), dict(output_domain="spectral", compute_stats=True),
               dict(mask=field > -15.0, normalize=True, compute_stats=True, subtract_mean=True)):
        want = ref_decomp(field, bp, fft_method="numpy", **kw)
        got = decomposition_fft(field, bp, **kw)
        assert set(got) == set(want)
        for a, b in zip(got["cascade_levels"], want["cascade_levels"]):
            assert np.array_equal(a, b)
        assert np.array_equal(got["means"], want["means"]) and np.array_equal(got["stds"], want["stds"])
    want = ref_decomp(field, bp, fft_method="numpy", normalize=True, compute_stats=True)
    assert np.array_equal(recompose_fft(want), ref_recomp(want))
    with pytest.raises(ValueError):
        decomposition_fft(field[:100], bp)  # the reference's dimension check, through the delegation


def test_noise_generator_on_other_shapes_is_the_reference(ref_pysteps):
    from pysteps.noise import fftgenerators as ref

    from pysteps_amd.noise import generate_noise_2d_fft_filter

    shape = (4100, 10)
    field = _field(shape, 2)
    pg = ref.initialize_nonparam_2d_fft_filter(field)
    for domain in ("spatial", "spectral"):
        want = ref.generate_noise_2d_fft_filter(pg, randstate=np.random.RandomState(5), domain=domain)
        got = generate_noise_2d_fft_filter(pg, randstate=np.random.RandomState(5), domain=domain)
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        generate_noise_2d_fft_filter(pg, domain="nowhere")


def test_method_tables_after_register(ref_pysteps):
    from pysteps import cascade, noise

    from pysteps_amd import register
    from pysteps_amd.cascade import decomposition_fft, recompose_fft
    from pysteps_amd.noise import generate_noise_2d_fft_filter

    try:
        register.register()
        assert cascade.get_method("fft_hip") == (decomposition_fft, recompose_fft)
        init, gen = noise.get_method("nonparametric_hip")
        assert gen is generate_noise_2d_fft_filter and init is noise.get_method("nonparametric")[0]
        assert noise.get_method("parametric_hip")[0] is noise.get_method("parametric")[0]
        assert cascade.get_method("fft")[0] is not decomposition_fft  # the stock names are untouched
    finally:
        register.unregister_fft()
