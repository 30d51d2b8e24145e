"""The algebra of the resident STEPS update on spectra (oracle/steps_spectral.py) against the chain of the reference's
spatial operators, in NumPy: even and odd grid sides, AR orders 1-3, two consecutive updates."""

import numpy as np
import pytest

from oracle import steps_spectral as oss


def _radial_filters(m, n, levels, rng):
    ky = np.fft.fftfreq(m)[:, None] * m
    kx = np.fft.rfftfreq(n)[None, :] * n
    r = np.hypot(ky, kx)
    centres = np.geomspace(1.0, max(m, n) / 2.5, levels)
    w = np.stack([np.exp(-0.5 * ((np.log(np.maximum(r, 0.5)) - np.log(c)) / 0.45) ** 2) for c in centres])
    w[0][r < centres[0]] = 1.0
    noise_filter = 1.0 / (1.0 + r) ** 1.3 * (1.0 + 0.2 * np.cos(r / 3.0))
    return w, noise_filter


@pytest.mark.parametrize("shape,p", [((32, 32), 2), ((30, 41), 2), ((33, 24), 1), ((21, 35), 3), ((64, 48), 2)])
def test_spectral_update_equals_the_spatial_chain(shape, p):
    m, n = shape
    L = 5
    rng = np.random.default_rng(m * 100 + n)
    weights, noise_filter = _radial_filters(m, n, L, rng)
    levels = rng.standard_normal((L, p, m, n))
    phi = rng.uniform(-0.8, 0.8, (L, p + 1))
    noise_std, mu, sigma = rng.uniform(0.3, 1.2, L), rng.standard_normal(L), rng.uniform(0.2, 2.0, L)
    spectra = np.fft.rfft2(levels)
    for step in range(2):
        white = rng.standard_normal((m, n))
        new_f, field_f = oss.update_spatial(white, noise_filter, weights, levels, phi, noise_std, mu, sigma)
        new_s, field_s = oss.update_spectral(white, noise_filter, weights, spectra, phi, noise_std, mu, sigma)
        scale = np.ptp(field_f)
        assert np.max(np.abs(field_s - field_f)) < 1e-12 * scale, (shape, step)
        assert np.max(np.abs(np.fft.irfft2(new_s, s=(m, n)) - new_f)) < 1e-12 * np.ptp(new_f)
        levels = np.concatenate([levels[:, 1:], new_f[:, None]], axis=1)
        spectra = np.concatenate([spectra[:, 1:], new_s[:, None]], axis=1)


def test_hermitian_weights_are_parsevals():
    rng = np.random.default_rng(1)
    for (m, n) in ((8, 8), (7, 9), (12, 5), (6, 10)):
        x = rng.standard_normal((m, n))
        spec = np.fft.rfft2(x)
        total = np.sum(oss.hermitian_weights(n)[None, :] * np.abs(spec) ** 2)
        assert abs(total - m * n * np.sum(x**2)) < 1e-9 * total
