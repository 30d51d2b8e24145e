"""TEST INFRASTRUCTURE ONLY (never imported by the product).

NumPy restatement of one STEPS member update in its two formulations, to pin the algebra of the resident update
(pysteps_amd/nowcasts/steps_resident.py, csrc/steps_loop.hip) on the CPU:

* ``update_spatial``  - the reference's chain of spatial operators: noise filter + standardisation
  (pysteps/noise/fftgenerators.py:400-437), band-pass decomposition with level statistics and normalisation
  (pysteps/cascade/decomposition.py:199-232), ``eps *= noise_std_coeffs`` and the AR(p) step
  (pysteps/nowcasts/steps.py:1131-1139, pysteps/timeseries/autoregression.py:1056-1070), recomposition
  (pysteps/cascade/decomposition.py:294-301);
* ``update_spectral`` - the same update with the AR history kept as spectra: level variances by Parseval's identity
  (Hermitian-weighted sums over the rfft2 half spectrum, DC excluded), ONE inverse transform.
Both return (new level state, recomposed field); tests/test_steps_spectral_cpu.py holds them equal to rounding for even
and odd grid sides.
* ``update_reference_spectral_domain`` - the member update of the reference's OWN ``domain="spectral"`` (a different
  state - compact spectral levels - and a different random stream - uniform phases), operation for operation; pinned
  against the real ``StepsNowcaster.__update_state`` (bit-identical) in the same test file.
"""

import numpy as np


def update_spatial(white, noise_filter, weights, levels, phi, noise_std, mu, sigma):
    """levels: (L, p, m, n) level fields, oldest first.  Returns (new fields (L, m, n), recomposed field)."""
    m, n = white.shape
    noise = np.fft.irfft2(np.fft.rfft2(white) * noise_filter, s=(m, n))
    noise = (noise - noise.mean()) / noise.std()
    spec = np.fft.rfft2(noise)
    new = []
    for k in range(weights.shape[0]):
        eps = np.fft.irfft2(spec * weights[k], s=(m, n))
        eps = (eps - eps.mean()) / eps.std()
        eps = eps * noise_std[k]
        p = levels.shape[1]
        x_new = 0.0
        for j in range(p):
            x_new = x_new + phi[k, j] * levels[k, p - 1 - j]
        new.append(x_new + phi[k, p] * eps)
    new = np.stack(new)
    field = np.sum(np.stack([new[k] * sigma[k] + mu[k] for k in range(len(new))]), axis=0)
    return new, field


def hermitian_weights(n):
    """Multiplicity of every column of an rfft2 half spectrum of a real field with n columns."""
    nc = n // 2 + 1
    w = np.full(nc, 2.0)
    w[0] = 1.0
    if n % 2 == 0:
        w[-1] = 1.0
    return w


def update_spectral(white, noise_filter, weights, spectra, phi, noise_std, mu, sigma):
    """spectra: (L, p, m, n/2+1) = rfft2 of the level fields, oldest first.  Returns (new spectra (L, m, n/2+1), field)."""
    m, n = white.shape
    y = np.fft.rfft2(white) * noise_filter
    y[0, 0] = 0.0  # the standardised noise field has no mean
    hw = hermitian_weights(n)[None, :]
    p = spectra.shape[1]
    total = np.zeros_like(y)
    new = []
    for k in range(weights.shape[0]):
        e = y * weights[k]
        b = float(np.sum(hw * np.abs(e) ** 2))
        gain = phi[k, p] * noise_std[k] * (m * n) / np.sqrt(b)
        x_new = gain * e
        for j in range(p):
            x_new = x_new + phi[k, j] * spectra[k, p - 1 - j]
        new.append(x_new)
        total = total + sigma[k] * x_new
    total[0, 0] += np.sum(mu) * m * n
    return np.stack(new), np.fft.irfft2(total, s=(m, n))


def spectral_std(x, shape):
    """pysteps/utils/spectral.py:208-238 for an rfft2 half spectrum."""
    m, n = shape
    res = np.sum(np.abs(x) ** 2) - np.real(x[0, 0]) ** 2
    res += np.sum(np.abs(x[:, 1:]) ** 2) if n % 2 == 1 else np.sum(np.abs(x[:, 1:-1]) ** 2)
    return np.sqrt(res / (m * n) ** 2)


def update_reference_spectral_domain(randstate, shape, noise_filter, weights, compact_levels, phi, noise_std, mu, sigma):
    """One member update of ``nowcasts.steps(domain="spectral")``, operation for operation:
    pysteps/noise/fftgenerators.py:407-437 (unit phasors from ``randstate.uniform``, column 0 mirrored, filter, DC
    removed, spectral standardisation), pysteps/cascade/decomposition.py:195-236 with spectral input and output,
    ``normalize`` and ``compact_output`` (level k = field[weights_k > 1e-12], mean and standard deviation from
    utils/spectral.py), pysteps/nowcasts/steps.py:1131-1146 + timeseries/autoregression.py:1060-1073 (AR step on the
    compact arrays), cascade/decomposition.py:284-300 (``result[mask_k] += level_k * sigma_k + mu_k``) and the one
    inverse transform of steps.py:1188-1189.
    compact_levels: list of L arrays (p, count_k) complex, oldest first (updated in place like the reference's
    ``np.concatenate``).  Returns the recomposed field (m, n)."""
    m, n = shape
    nc = n // 2 + 1
    theta = randstate.uniform(low=0.0, high=2.0 * np.pi, size=(m, nc))
    half = m // 2
    if m % 2 == 0:
        theta[half + 1:, 0] = -theta[1:half, 0][::-1]
    else:
        theta[half + 1:, 0] = -theta[1:half + 1, 0][::-1]
    noise = np.cos(theta) + 1.0j * np.sin(theta)
    noise *= noise_filter
    noise[0, 0] = 0.0
    noise /= spectral_std(noise, shape)
    result = np.zeros((m, nc), dtype=complex)
    p = phi.shape[1] - 1
    for k in range(weights.shape[0]):
        level = noise * weights[k]
        mean = np.real(level[0, 0]) / (m * n)
        std = spectral_std(level, shape)
        level = (level - mean) / std
        mask = weights[k] > 1e-12
        eps = level[mask]
        eps *= noise_std[k]
        x = compact_levels[k]
        x_new = 0.0
        for i in range(p):
            x_new = x_new + phi[k, i] * x[-(i + 1), :]
        x_new = x_new + phi[k, -1] * eps
        compact_levels[k] = np.concatenate([x[1:, :], x_new[np.newaxis, :]])
        result[mask] += x_new * sigma[k] + mu[k]
    return np.fft.irfft2(result, s=shape)
