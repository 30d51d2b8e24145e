"""ORACLE (test infrastructure only - never imported by pysteps_amd): NumPy's legacy
``RandomState.randn`` restated on top of NumPy's own uniform stream, with a correctly rounded
logarithm.

Follows numpy/random/src/legacy/legacy-distributions.c ``legacy_gauss`` (third party, numpy 2.2.6
installed; not under /root/reference): polar method on pairs of ``legacy_double`` values, the second
value of every pair cached.  ``RandomState.random_sample`` IS ``legacy_double`` (two MT19937 words,
``(a >> 5, b >> 6)`` -> 53 bits), so words, positions and accept / reject decisions come from NumPy
itself; the only thing restated is ``f = sqrt(-2 log(r2) / r2)``, with ``log`` evaluated by
``decimal`` at 60 digits and rounded once - the logarithm csrc/cr_log.h promises.  Pinned by
tests/test_rng_cpu.py: with ``log=numpy.log`` (the C library's) the restatement reproduces
``RandomState.randn`` bit for bit, state included.
"""

from decimal import Decimal, getcontext

import numpy as np


def log_correctly_rounded(x):
    getcontext().prec = 60
    return np.array([float(Decimal(float(v)).ln()) for v in np.ravel(x)]).reshape(np.shape(x))


def legacy_randn(randstate, count, log=log_correctly_rounded):
    """``count`` values of ``randstate.randn`` (the generator is advanced exactly as NumPy would
    advance it, cached value included)."""
    name, key, pos, has_gauss, cached = randstate.get_state(legacy=True)
    out = np.empty(count, dtype=np.float64)
    filled = 0
    if has_gauss and count > 0:
        out[0] = cached
        filled, has_gauss, cached = 1, 0, 0.0
    pairs = (count - filled + 1) // 2
    xs1, xs2, r2s = [], [], []
    remaining = pairs
    while remaining > 0:
        attempts = int(remaining / 0.785 * 1.02) + 16
        snapshot = randstate.get_state()
        u = randstate.random_sample(2 * attempts).reshape(attempts, 2)
        x1 = 2.0 * u[:, 0] - 1.0
        x2 = 2.0 * u[:, 1] - 1.0
        r2 = x1 * x1 + x2 * x2
        ok = ~((r2 >= 1.0) | (r2 == 0.0))
        cum = np.cumsum(ok)
        if cum[-1] >= remaining:  # stop right behind the attempt that completes the draw
            last = int(np.searchsorted(cum, remaining))
            randstate.set_state(snapshot)
            randstate.random_sample(2 * (last + 1))
            ok[last + 1:] = False
            remaining = 0
        else:
            remaining -= int(cum[-1])
        xs1.append(x1[ok]); xs2.append(x2[ok]); r2s.append(r2[ok])
    if pairs:
        x1, x2, r2 = np.concatenate(xs1), np.concatenate(xs2), np.concatenate(r2s)
        f = np.sqrt(-2.0 * log(r2) / r2)
        vals = np.empty(2 * pairs)
        vals[0::2] = f * x2
        vals[1::2] = f * x1
        n_out = count - filled
        out[filled:] = vals[:n_out]
        if n_out < 2 * pairs:
            has_gauss, cached = 1, float(vals[-1])
    st = randstate.get_state(legacy=True)
    randstate.set_state((st[0], st[1], st[2], has_gauss, cached))
    return out
