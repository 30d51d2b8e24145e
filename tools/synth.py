"""Seeded synthetic inputs shared by the golden generator, tests and bench.py.

Follows the recipe fixed in BASELINE.md section 3 / SURVEY.md section 8d: Gaussian-filtered noise
-> intermittent rain rate -> dB field, smooth sinusoidal true motion.
"""

import numpy as np


def rain_field_db(m, n, seed=1234, sigma=None):
    """(m,n) float32 dB rain field, about 30 % wet, zero value -15 dB."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(seed)
    sigma = m / 128.0 if sigma is None else sigma
    g = gaussian_filter(rng.standard_normal((m, n)), sigma=max(sigma, 1.0))
    rate = np.maximum(g / g.std() * 8.0 - 4.0, 0.0)
    with np.errstate(divide="ignore"):
        db = np.where(rate > 0.1, 10.0 * np.log10(np.maximum(rate, 1e-30)), -15.0)
    return db.astype(np.float32)


def true_velocity(m, n, dtype=np.float32):
    """(2,m,n) [0]=u along x, [1]=v along y, px/step, |V| <= 7.5."""
    y, x = np.mgrid[0:m, 0:n].astype(np.float64)
    u = 4.0 + 2.0 * np.sin(2.0 * np.pi * y / m)
    v = -3.0 + 1.5 * np.cos(2.0 * np.pi * x / n)
    return np.stack([u, v]).astype(dtype)


def border_nan_mask(m, n, frac=0.1):
    """Boolean mask that is True on a ragged `frac` border (radar-mask look-alike)."""
    y, x = np.mgrid[0:m, 0:n]
    cy, cx = (m - 1) / 2.0, (n - 1) / 2.0
    r = np.hypot((y - cy) / (m / 2.0), (x - cx) / (n / 2.0))
    return r > (1.0 - frac) * np.sqrt(2.0) * 0.75


def steps_frames(m, n, n_frames=3, seed=1234, advect=None):
    """(n_frames,m,n) float32 dB frames for STEPS-like callers (SURVEY.md section 8d, config 4 caveat):
    frame t = the base field advected t steps by ``true_velocity`` PLUS temporal evolution
    ``where(adv>-15, max(adv + e_t, -10), -15)`` with e_t a unit-variance smooth field (seed 100+t) --
    pure advections of one field have Lagrangian lag-correlation 1 and make the Yule-Walker fit of
    STEPS singular.  ``advect(P, V, t)`` -> advected field (default: np.roll by the mean motion,
    which needs neither the oracle nor the reference)."""
    from scipy.ndimage import gaussian_filter

    base = rain_field_db(m, n, seed=seed, sigma=max(m / 64.0, 2.0))
    vel = true_velocity(m, n)
    frames = []
    for t in range(n_frames):
        if advect is None:
            adv = np.roll(base, (int(round(-3.0 * t)), int(round(4.0 * t))), axis=(0, 1))
        else:
            adv = base if t == 0 else advect(base, vel, t)
        rng = np.random.default_rng(100 + t)
        e = gaussian_filter(rng.standard_normal((m, n)), max(m / 256.0, 1.5))
        e /= e.std()
        frames.append(np.where(adv > -15.0, np.maximum(adv + 1.0 * e, -10.0), -15.0))
    return np.stack(frames).astype(np.float32)
