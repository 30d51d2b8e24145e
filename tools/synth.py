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
