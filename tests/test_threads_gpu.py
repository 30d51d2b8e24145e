"""The reference operators are re-entrant and are invoked from dask threads
(pysteps/nowcasts/utils.py:464-468, steps.py:710-720); the drop-in has to be thread-safe
(SURVEY 8b "Threading").  Concurrent callers must get exactly what sequential callers get."""

from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields(seed, m=96, n=128):
    from tools import synth

    p = synth.rain_field_db(m, n, seed=seed, sigma=3.0)
    y, x = np.mgrid[0:m, 0:n]
    v = synth.true_velocity(m, n) + np.stack([0.01 * seed * (x - n / 2), -0.02 * (y - m / 2)]).astype(np.float32)
    return p, v


def test_extrapolate_from_threads():
    from pysteps_amd.extrapolation import get_method

    ex = get_method("semilagrangian")
    jobs = [(_fields(s), dict(n_iter=s % 3, interp_order=(0, 1, 3)[s % 3], outval=-15.0)) for s in range(12)]
    want = [ex(p, v, 3, return_displacement=True, **kw) for (p, v), kw in jobs]
    with ThreadPoolExecutor(max_workers=6) as pool:
        got = list(pool.map(lambda j: ex(j[0][0], j[0][1], 3, return_displacement=True, **j[1]), jobs * 3))
    for i, (out, disp) in enumerate(got):
        w_out, w_disp = want[i % len(jobs)]
        assert np.array_equal(out, w_out, equal_nan=True) and np.array_equal(disp, w_disp)


def test_dense_lk_and_member_steps_from_threads():
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from pysteps_amd.motion import get_method

    lk = get_method("LK")
    frames = []
    for s in range(4):
        p, _ = _fields(20 + s, 160, 192)
        frames.append(np.stack([p, np.roll(p, (1 + s % 2, 2), axis=(0, 1))]))
    want = [lk(f) for f in frames]

    def member_step(seed):
        p, v = _fields(seed)
        adv = EnsembleAdvector(v, 2, n_iter=1)
        out = adv.step(np.stack([p, p + 1.0]).astype(np.float32), [1.0, 1.0])
        return out, adv.displacement.to_host()

    want_members = [member_step(s) for s in range(4)]
    with ThreadPoolExecutor(max_workers=8) as pool:
        f_lk = [pool.submit(lk, f) for f in frames * 2]
        f_mb = [pool.submit(member_step, s) for s in list(range(4)) * 2]
        got_lk = [f.result() for f in f_lk]
        got_mb = [f.result() for f in f_mb]
    for i, g in enumerate(got_lk):
        assert np.array_equal(g, want[i % 4])
    for i, (o, d) in enumerate(got_mb):
        assert np.array_equal(o, want_members[i % 4][0], equal_nan=True) and np.array_equal(d, want_members[i % 4][1])
