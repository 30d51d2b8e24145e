"""HIP FFTs (csrc/fft.hip, pysteps_amd/utils/fft.py) against numpy.fft - the reference's default FFT
method (pysteps/utils/fft.py:20-37) - and through the reference's own callers.

numpy's transforms are pocketfft in float64 [third party, numpy 2.2, installed]; the HIP
kernels compute the same mathematical transform in float64 with a different factorisation, so
parity is to round-off: rel-L2 <= 1e-12 (observed ~1e-16 x log2 size).
"""

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-12
SHAPES = [(2, 2), (2, 8), (4, 4), (16, 2), (8, 32), (64, 64), (32, 128), (256, 512), (1024, 256), (2048, 2048)]
# sides that are not powers of two (Bluestein inside the same kernels): odd and even lengths, primes,
# one plain and one chirp-z axis, and the radar composites the reference's examples use
ANY_SHAPES = [(3, 5), (7, 2), (2, 7), (6, 10), (17, 31), (100, 64), (64, 100), (127, 257), (640, 710), (1226, 760),
              (4096, 1000), (3000, 2048), (4095, 6)]


def _c(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def test_irfft2_of_non_hermitian_spectra_any_length():
    """odd lengths have no Nyquist bin: numpy drops the imaginary part of bin 0 only"""
    from pysteps_amd.utils.fft import get_hip

    rng = np.random.default_rng(9)
    for shape in ((33, 45), (40, 54), (45, 64)):
        X = rng.standard_normal((shape[0], shape[1] // 2 + 1)) + 1j * rng.standard_normal((shape[0], shape[1] // 2 + 1))
        assert _c(get_hip(shape).irfft2(X), np.fft.irfft2(X, s=shape)) < TOL


@pytest.mark.parametrize("shape", SHAPES + ANY_SHAPES)
def test_transforms_match_numpy(shape):
    from pysteps_amd.utils.fft import get_hip

    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    fft = get_hip(shape)
    x = rng.standard_normal(shape) * 3 + 1.5
    got = fft.rfft2(x)
    want = np.fft.rfft2(x)
    assert got.shape == want.shape and got.dtype == np.complex128
    assert _c(got, want) < TOL
    back = fft.irfft2(want)
    assert back.shape == shape and back.dtype == np.float64
    assert _c(back, np.fft.irfft2(want, s=shape)) < TOL and _c(back, x) < TOL
    z = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    assert _c(fft.fft2(z), np.fft.fft2(z)) < TOL
    assert _c(fft.ifft2(z), np.fft.ifft2(z)) < TOL
    assert _c(fft.fft2(x), np.fft.fft2(x)) < TOL  # real input
    # single precision in -> single precision out, as numpy >= 2.0 (which also COMPUTES in float32:
    # the bar here is float32 round-off)
    x32 = x.astype(np.float32)
    got32, want32 = fft.rfft2(x32), np.fft.rfft2(x32)
    assert got32.dtype == want32.dtype and _c(got32, want32) < 1e-5
    assert fft.irfft2(want32).dtype == np.fft.irfft2(want32, s=shape).dtype
    assert fft.fft2(x32).dtype == np.fft.fft2(x32).dtype
    assert _c(fft.rfft2(x.astype(np.int32)), np.fft.rfft2(x.astype(np.int32))) < TOL


def test_irfft2_ignores_the_imaginary_parts_numpy_ignores():
    """A spectrum that is not Hermitian (STEPS multiplies spectra by real filters, but the property
    is numpy's): irfft2 drops the imaginary parts of the zero / Nyquist bins of the last axis."""
    from pysteps_amd.utils.fft import get_hip

    shape = (64, 128)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((64, 65)) + 1j * rng.standard_normal((64, 65))
    assert _c(get_hip(shape).irfft2(X), np.fft.irfft2(X, s=shape)) < TOL


def test_full_size_and_resident_arrays():
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.utils.fft import get_hip

    for shape in ((4096, 4096), (512, 8192), (8192, 1024)):
        rng = np.random.default_rng(shape[1])
        x = rng.standard_normal(shape)
        fft = get_hip(shape)
        want = np.fft.rfft2(x)
        dx = DeviceArray.from_host(x)
        dX = fft.rfft2(dx)  # resident in, resident out
        assert isinstance(dX, DeviceArray) and dX.shape == want.shape and dX.dtype == np.complex128
        assert _c(dX.to_host(), want) < TOL
        back = fft.irfft2(dX)
        assert isinstance(back, DeviceArray) and _c(back.to_host(), x) < TOL
        assert _c(dX.to_host(), want) < TOL  # the inverse left its input untouched
        # linearity and Parseval at full size
        y = rng.standard_normal(shape)
        lhs = fft.rfft2(2.0 * x - 0.5 * y)
        assert _c(lhs, 2.0 * want - 0.5 * fft.rfft2(y)) < TOL
        full = fft.fft2(x)
        assert abs(np.sum(np.abs(full) ** 2) / x.size - np.sum(x * x)) < 1e-10 * np.sum(x * x)


def test_reference_noise_and_cascade_callers(ref_pysteps):
    """pysteps.noise.fftgenerators and pysteps.cascade.decomposition with the HIP method object vs
    the numpy method object (the calls of the STEPS member loop, steps.py:1147-1171)."""
    from pysteps import utils
    from pysteps.cascade.bandpass_filters import filter_gaussian
    from pysteps.cascade.decomposition import decomposition_fft, recompose_fft
    from pysteps.noise import fftgenerators

    from pysteps_amd import register
    from tools import synth

    register.register()
    shape = (256, 256)
    field = synth.rain_field_db(*shape, seed=9).astype(np.float64)
    hip = utils.get_method("hip", shape=shape)
    ref = utils.get_method("numpy", shape=shape)
    pg_h = fftgenerators.initialize_param_2d_fft_filter(field, fft_method=hip)
    pg_r = fftgenerators.initialize_param_2d_fft_filter(field, fft_method=ref)
    # (the default "power-law" model FITS a parametric spectrum to the transform with scipy's
    # curve_fit: round-off differences of the transform are amplified by the optimiser)
    assert _c(pg_h["field"], pg_r["field"]) < 1e-6
    np_h = fftgenerators.initialize_nonparam_2d_fft_filter(field, fft_method=hip)
    np_r = fftgenerators.initialize_nonparam_2d_fft_filter(field, fft_method=ref)
    assert _c(np_h["field"], np_r["field"]) < 1e-10
    n_h = fftgenerators.generate_noise_2d_fft_filter(pg_r, randstate=np.random.RandomState(3), fft_method=hip)
    n_r = fftgenerators.generate_noise_2d_fft_filter(pg_r, randstate=np.random.RandomState(3), fft_method=ref)
    assert _c(n_h, n_r) < 1e-10
    bp = filter_gaussian(shape, 6)
    for domain in ("spatial", "spectral"):
        d_h = decomposition_fft(field, bp, fft_method=hip, output_domain=domain, normalize=True, compute_stats=True,
                                compact_output=domain == "spectral")
        d_r = decomposition_fft(field, bp, fft_method=ref, output_domain=domain, normalize=True, compute_stats=True,
                                compact_output=domain == "spectral")
        for a, b in zip(d_h["cascade_levels"], d_r["cascade_levels"]):
            assert _c(np.asarray(a), np.asarray(b)) < 1e-10
        assert np.allclose(d_h["means"], d_r["means"], rtol=1e-10, atol=1e-12)
        assert np.allclose(d_h["stds"], d_r["stds"], rtol=1e-10)
    assert _c(recompose_fft(d_h), recompose_fft(d_r)) < 1e-10
    # sides that are not powers of two run on the device too (chirp-z), longer ones go to numpy.fft unchanged
    odd = utils.get_method("hip", shape=(200, 200))
    y = np.random.default_rng(1).standard_normal((200, 200))
    assert _c(odd.rfft2(y), np.fft.rfft2(y)) < TOL and not np.array_equal(odd.rfft2(y), np.fft.rfft2(y))
    long_side = utils.get_method("hip", shape=(5000, 6))
    y = np.random.default_rng(2).standard_normal((5000, 6))
    assert np.array_equal(long_side.rfft2(y), np.fft.rfft2(y))


def test_nowcasts_steps_with_the_hip_fft_method(ref_pysteps):
    """nowcasts.steps picks the method up by name (steps.py:637, 1008): fft_method="hip" vs "numpy",
    same seed.  The member loop thresholds and rank-matches the fields, so a round-off difference may
    move isolated pixels across a threshold: all but 1e-3 of the pixels agree to 1e-6."""
    from pysteps import nowcasts

    from pysteps_amd import register
    from tools import synth

    register.register()
    frames = synth.steps_frames(256, 256, 3)
    V = synth.true_velocity(256, 256).astype(np.float64)
    steps = nowcasts.get_method("steps")
    kw = dict(n_ens_members=4, n_cascade_levels=6, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, seed=42,
              vel_pert_method="bps", mask_method="incremental", num_workers=1)
    want = steps(frames, V, 3, extrap_method="semilagrangian", fft_method="numpy", **kw)
    got = steps(frames, V, 3, extrap_method="semilagrangian", fft_method="hip", **kw)
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = np.isfinite(want)
    far = np.abs(got[ok] - want[ok]) > 1e-6 * (1.0 + np.abs(want[ok]))
    assert far.mean() < 1e-3, far.mean()
