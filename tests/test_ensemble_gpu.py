"""Member-batched stateful advection vs per-member oracle calls (nowcasts/utils.py:441-462 semantics)."""

import numpy as np
import pytest

from conftest import nan_mismatch, rel_l2

pytestmark = pytest.mark.gpu


def _perturbators(n, seed):
    rng = np.random.default_rng(seed)
    # parameters of pysteps.noise.motion.initialize_bps (defaults p_par/p_perp of the reference)
    return [dict(eps_par=rng.laplace(scale=1 / np.sqrt(2)), eps_perp=rng.laplace(scale=1 / np.sqrt(2)),
                 p_par=(10.88, 0.23, -7.68), p_perp=(5.76, 0.31, -2.72), vsf=60.0 / (5.0 * 1.0)) for _ in range(n)]


def _generate_bps(V, p, t):
    """NumPy restatement of noise/motion.py:127-131,146-180 for the oracle side."""
    N = np.linalg.norm(V, axis=0)
    Vn = np.where(N > 1e-12, V / np.where(N > 1e-12, N, 1.0), 0.0)
    Vp = np.stack([-Vn[1], Vn[0]])
    g_par = p["p_par"][0] * pow(t, p["p_par"][1]) + p["p_par"][2]
    g_perp = p["p_perp"][0] * pow(t, p["p_perp"][1]) + p["p_perp"][2]
    return (g_par * p["eps_par"] * Vn + g_perp * p["eps_perp"] * Vp) / p["vsf"]


@pytest.mark.parametrize("perturb,n_iter", [(False, 1), (True, 1), (True, 0), (True, 3)])
def test_members_match_per_member_oracle(perturb, n_iter):
    from oracle import semilag as osl
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    B, m, n = 5, 96, 128
    members = np.stack([synth.rain_field_db(m, n, seed=40 + j, sigma=2.0) for j in range(B)])
    members[2, 10:20, 30:40] = np.nan
    V = synth.true_velocity(m, n)
    V[:, 5, 7] = 0.0  # a calm pixel: V_par must be zero there
    perts = _perturbators(B, 3) if perturb else None
    adv = EnsembleAdvector(V, B, perts, n_iter=n_iter)
    D = [None] * B
    for step, (dt, t_total) in enumerate([(1.0, 5.0), (1.0, 10.0), (0.5, 12.5)]):
        got = adv.step(members, dt, t_total)
        assert got.shape == (B, m, n)
        for j in range(B):
            Vj = V.astype(np.float64) + (_generate_bps(V.astype(np.float64), perts[j], t_total) if perturb else 0.0)
            want, D[j] = osl.extrapolate(members[j], Vj, [dt], allow_nonfinite_values=True, n_iter=n_iter,
                                         return_displacement=True, displacement_prev=D[j])
            assert nan_mismatch(got[j], want[0]) <= 2  # NaN edge can move by a pixel at exact integers
            assert rel_l2(got[j], want[0]) < 1e-4
    gd = adv.displacement.to_host()
    for j in range(B):
        assert np.max(np.abs(gd[j] - D[j])) < 1e-4


def test_members_4096_config4_vs_per_member_oracle():
    """BASELINE config 4's resident member advection at full size (4096^2, perturbed motion, two members,
    two time steps with the displacement carried over) against per-member calls of the SciPy oracle;
    the bars are the regression bars of the small cases, what was seen goes to gpurun_out/."""
    from oracle import semilag as osl
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    B, m, n = 2, 4096, 4096
    members = np.stack([synth.rain_field_db(m, n, seed=70 + j) for j in range(B)])
    members[1, 100:140, 3000:3100] = np.nan
    V = synth.true_velocity(m, n)
    perts = _perturbators(B, 11)
    adv = EnsembleAdvector(V, B, perts, n_iter=1)
    D = [None] * B
    seen = []
    for dt, t_total in [(1.0, 5.0), (1.0, 10.0)]:
        got = adv.step(members, dt, t_total)
        for j in range(B):
            Vj = V.astype(np.float64) + _generate_bps(V.astype(np.float64), perts[j], t_total)
            want, D[j] = osl.extrapolate(members[j], Vj, [dt], allow_nonfinite_values=True, n_iter=1,
                                         return_displacement=True, displacement_prev=D[j])
            assert nan_mismatch(got[j], want[0]) <= 8
            seen.append(rel_l2(got[j], want[0]))
            assert seen[-1] < 5e-7  # observed 3.4e-8 .. 3.7e-8 (profiles/r05/members_4096_seen.json); the contract is 1e-4
    gd = adv.displacement.to_host()
    dmax = max(float(np.max(np.abs(gd[j] - D[j]))) for j in range(B))
    assert dmax < 1e-5  # observed 2.0e-6
    try:
        import json
        import os

        os.makedirs("gpurun_out", exist_ok=True)
        with open("gpurun_out/members_4096_seen.json", "w") as fh:
            json.dump({"rel_l2": seen, "disp_max_abs": dmax}, fh)
    except OSError:
        pass


def test_members_equal_single_member_calls():
    """Without perturbation the batched kernel reproduces the fused single-field kernel."""
    from pysteps_amd.extrapolation import get_method
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    B, m, n = 3, 200, 192
    members = np.stack([synth.rain_field_db(m, n, seed=j) for j in range(B)])
    V = synth.true_velocity(m, n)
    adv = EnsembleAdvector(V, B, outval=-15.0)
    ex = get_method("semilagrangian")
    d = [None] * B
    for _ in range(3):
        got = adv.step(members, 1.0)
        for j in range(B):
            want, d[j] = ex(members[j], V, [1.0], outval=-15.0, return_displacement=True, displacement_prev=d[j])
            assert np.max(np.abs(got[j] - want[0])) < 1e-5
    assert adv.step(None, 1.0) is None  # displacement-only call (utils.py:498-503)


def test_single_member_and_nearest_order():
    from pysteps_amd.extrapolation import get_method
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    m, n = 70, 130
    p = synth.rain_field_db(m, n, seed=2)[None]
    V = synth.true_velocity(m, n)
    adv = EnsembleAdvector(V, 1, interp_order=0, outval=-15.0)
    got = adv.step(p, [1.0, 1.0])  # two lead-time INCREMENTS in one call -> (B, T, m, n)
    want = get_method("semilagrangian")(p[0], V, [1.0, 2.0], outval=-15.0, interp_order=0)
    assert got.shape == (1, 2, m, n)
    assert np.count_nonzero(got[0] != want) <= 1e-4 * want.size


def test_compact_state_float64_entry_and_restart():
    """The resident 16-byte trajectory records and the float64 displacement of the reference are
    two views of the same state: the float64 entry point (psh_semilag_members_dev) continues from
    ``adv.displacement`` exactly like the advector itself, and an advector restarted from a
    displacement array reproduces the original one."""
    from pysteps_amd import _lib
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    B, m, n = 3, 70, 200
    members = np.stack([synth.rain_field_db(m, n, seed=60 + j, sigma=2.0) for j in range(B)])
    V = synth.true_velocity(m, n) * 2.5
    adv = EnsembleAdvector(V, B, n_iter=1)
    assert not adv.displacement.to_host().any()  # zeros before the first step
    adv.step(members, [1.0, 1.0])
    disp = adv.displacement  # float64 (B,2,m,n) on the device
    assert disp.dtype == np.float64 and disp.shape == (B, 2, m, n)
    d_host = disp.to_host()
    # the records hold D = (P - x) + frac with frac in [0, 1)
    assert np.all(d_host - np.floor(d_host) < 1.0)
    want = adv.step(members, 0.5)
    # (a) float64 entry point, continuing from the converted displacement
    lib = _lib.lib()
    pm, dv = DeviceArray.from_host(members, np.float32), DeviceArray.from_host(V, np.float32)
    out = DeviceArray((B, 1, m, n), np.float32)
    steps = np.array([0.5])
    _lib.check(lib.psh_semilag_members_dev(pm.ptr, dv.ptr, None, None, None, B, m, n, steps.ctypes.data, 1, 1, 1,
                                           float("nan"), disp.ptr, 1, out.ptr), "psh_semilag_members_dev")
    assert np.array_equal(out.to_host()[:, 0], want, equal_nan=True)
    assert np.array_equal(disp.to_host(), adv.displacement.to_host())
    # (b) restart from a host displacement array
    again = EnsembleAdvector(V, B, n_iter=1)
    again.displacement = d_host
    assert np.array_equal(again.step(members, 0.5), want, equal_nan=True)
    with pytest.raises(ValueError):
        again.displacement = d_host[:, :1]


@pytest.mark.parametrize("perturb,n_iter,order", [(True, 1, 1), (False, 1, 1), (True, 3, 1), (True, 0, 0), (False, 2, 0)])
def test_packed_gather_planes_are_bit_identical(perturb, n_iter, order):
    """The interleaved-plane kernels (one dwordx4 per tap) against the one-plane-per-component ones."""
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    B, m, n = 3, 150, 257  # odd width: border waves, unaligned rows
    members = np.stack([synth.rain_field_db(m, n, seed=80 + j, sigma=2.0) for j in range(B)])
    members[1, 40:50, 100:120] = np.nan
    V = synth.true_velocity(m, n) * 1.7
    V[:, 7, 9] = 0.0
    perts = _perturbators(B, 11) if perturb else None
    a = EnsembleAdvector(V, B, perts, n_iter=n_iter, interp_order=order, outval=-15.0, packed=True)
    b = EnsembleAdvector(V, B, perts, n_iter=n_iter, interp_order=order, outval=-15.0, packed=False)
    assert a.packed is not None and b.packed is None
    for dt, t_total in [(1.0, 5.0), (0.5, 7.5), (2.0, 17.5)]:
        ga, gb = a.step(members, dt, t_total), b.step(members, dt, t_total)
        assert np.array_equal(ga, gb, equal_nan=True)
    assert np.array_equal(a.displacement.to_host(), b.displacement.to_host())


@pytest.mark.parametrize("B,order,n_iter", [(3, 1, 1), (4, 1, 2), (5, 0, 1), (2, 1, 1)])
def test_two_members_per_thread_is_bit_identical(B, order, n_iter):
    """members_variant 2 (the default: a thread carries the same pixel of two members so that the two
    chains of gathers overlap) against variant 1 (one member per thread): the same arithmetic per
    trajectory - outputs and trajectory records bit for bit, odd member counts, image borders, a
    zero-velocity pixel, displacement-only calls in between"""
    from pysteps_amd import _lib
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    m, n = 130, 200
    members = np.stack([synth.rain_field_db(m, n, seed=40 + j, sigma=2.0) for j in range(B)])
    members[0, 5:9, 7:30] = np.nan
    V = synth.true_velocity(m, n)
    V[:, 17, 23] = 0.0
    perts = [dict(eps_par=0.8 - 0.5 * j, eps_perp=-0.4 + 0.3 * j, p_par=(10.88, 0.23, -7.68), p_perp=(5.76, 0.31, -2.72), vsf=12.0)
             for j in range(B)]
    got = {}
    try:
        for variant in (1, 2):
            _lib.check(_lib.lib().psh_set_option(b"members_variant", variant))
            adv = EnsembleAdvector(V, B, perts, outval=-15.0, n_iter=n_iter, interp_order=order)
            outs = [adv.step(members, 1.0, 5.0), adv.step(members, [1.0, 0.5], 10.0)]
            adv.step(None, 1.0, 15.0)
            outs.append(adv.step(members, 2.0, 20.0))
            got[variant] = (outs, adv._state.to_host())
    finally:
        _lib.check(_lib.lib().psh_set_option(b"members_variant", 2))
    np.testing.assert_array_equal(got[1][1], got[2][1])
    for a, b in zip(got[1][0], got[2][0]):
        np.testing.assert_array_equal(a, b)
