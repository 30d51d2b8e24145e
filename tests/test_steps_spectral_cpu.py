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


def test_filters_that_break_the_hermitian_symmetry_are_recognised():
    """An asymmetric value on a self-conjugate column of the half spectrum breaks the equivalence (1e-4 here); the
    resident update recognises such filters (steps_resident._self_conjugate_columns_symmetric) and keeps the spatial chain."""
    sr = pytest.importorskip("pysteps_amd.nowcasts.steps_resident")
    rng = np.random.default_rng(0)
    for (m, n) in ((32, 32), (30, 41), (33, 24)):
        w, f = _radial_filters(m, n, 4, rng)
        assert sr._self_conjugate_columns_symmetric(w, n) and sr._self_conjugate_columns_symmetric(f, n)
        spec = np.abs(np.fft.rfft2(rng.standard_normal((m, n))))  # what initialize_nonparam_2d_fft_filter keeps: symmetric to rounding
        assert sr._self_conjugate_columns_symmetric(spec, n)
        bad = w.copy()
        bad[1, 3, 0] *= 1.7
        assert not sr._self_conjugate_columns_symmetric(bad, n)
        if n % 2 == 0:
            bad = f.copy()
            bad[5, -1] += 0.3
            assert not sr._self_conjugate_columns_symmetric(bad, n)
        assert not sr._self_conjugate_columns_symmetric(w[..., :-1], n)  # not a half spectrum of this grid
    m, n, L, p = 32, 32, 4, 2
    w, f = _radial_filters(m, n, L, rng)
    w[:, 3, 0] *= 1.7
    levels = rng.standard_normal((L, p, m, n))
    phi = rng.uniform(-0.8, 0.8, (L, p + 1))
    ns, mu, sg = rng.uniform(0.3, 1.2, L), rng.standard_normal(L), rng.uniform(0.2, 2.0, L)
    white = rng.standard_normal((m, n))
    _, fa = oss.update_spatial(white, f, w, levels, phi, ns, mu, sg)
    _, fb = oss.update_spectral(white, f, w, np.fft.rfft2(levels), phi, ns, mu, sg)
    assert np.max(np.abs(fa - fb)) > 1e-6 * np.ptp(fa)


@pytest.mark.parametrize("shape", [(64, 64), (37, 53), (48, 31)])
def test_spectral_std_of_unit_phasor_noise_is_a_constant_of_the_filters(ref_pysteps, shape):
    """The reference's domain="spectral" noise is exp(i theta) x filter (fftgenerators.py:407-437): its spectral
    standard deviation (utils/spectral.py:208-238), and that of every cascade level, does not depend on the phases.
    `_spectral_std_of_moduli` - what the resident update computes once per nowcast - equals the reference's function
    on the complex field for any phases."""
    from pysteps.utils import spectral

    from pysteps_amd.nowcasts.steps_resident import _spectral_std_of_moduli

    m, n = shape
    rng = np.random.default_rng(5)
    filt = rng.uniform(0.0, 3.0, (m, n // 2 + 1))
    weights = rng.uniform(0.0, 1.0, (m, n // 2 + 1))
    filt[0, 0] = 0.0
    want_level = None
    for _ in range(3):
        theta = rng.uniform(0.0, 2.0 * np.pi, filt.shape)
        noise = (np.cos(theta) + 1j * np.sin(theta)) * filt
        got = _spectral_std_of_moduli(filt, m, n)
        assert abs(got - spectral.std(noise, (m, n))) <= 1e-14 * got
        level = spectral.std(noise / spectral.std(noise, (m, n)) * weights, (m, n))
        mine = _spectral_std_of_moduli(filt * (1.0 / got) * weights, m, n)
        assert abs(mine - level) <= 1e-14 * level
        want_level = level if want_level is None else want_level
        assert abs(level - want_level) <= 1e-14 * level


def test_restatement_of_the_reference_spectral_domain_update_is_the_reference(ref_pysteps):
    """oracle.steps_spectral.update_reference_spectral_domain against the REAL StepsNowcaster.__update_state of
    nowcasts.steps(domain="spectral"): no mask, no probability matching, so that the fields the function returns are
    the recomposed fields - identical to the last bit (same NumPy operations in the same order), and the generator
    ends where the reference's ends."""
    import copy

    from pysteps import nowcasts
    from pysteps.nowcasts import steps as steps_mod
    from tools import synth

    m, n = 48, 40
    frames = synth.steps_frames(m, n, 3)
    V = synth.true_velocity(m, n).astype(np.float64)
    seen = []

    def side_by_side(precip, velocity, state, timesteps, extrap_method, func, params=None, num_ensemble_members=1, **_):
        L = params["n_cascade_levels"]
        gens = [copy.deepcopy(g) for g in state["randgen_prec"]]
        levels = [[np.array(c, dtype=complex) for c in state["precip_cascades"][j]] for j in range(num_ensemble_members)]
        weights = np.asarray(params["filter"]["weights_2d"])
        phi = np.asarray(params["phi"])
        for _t in range(3):
            want, state = func(state, params)
            for j in range(num_ensemble_members):
                d = state["precip_decomp"][j]
                got = oss.update_reference_spectral_domain(gens[j], precip.shape, params["pert_gen"]["field"], weights, levels[j], phi,
                                                           np.asarray(params["noise_std_coeffs"]), d["means"], d["stds"])
                got[params["domain_mask"]] = np.nan
                seen.append(float(np.nanmax(np.abs(got - want[j]))))
                assert np.array_equal(np.isnan(got), np.isnan(want[j]))
                for k in range(L):
                    assert np.array_equal(levels[j][k], state["precip_cascades"][j][k])
        for g, h in zip(gens, state["randgen_prec"]):
            assert g.randint(0, 1 << 30) == h.randint(0, 1 << 30)
        return np.zeros((num_ensemble_members, timesteps) + precip.shape)

    orig = steps_mod.nowcast_main_loop
    try:
        steps_mod.nowcast_main_loop = side_by_side
        nowcasts.get_method("steps")(frames, V, 2, n_ens_members=2, n_cascade_levels=5, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0,
                                     seed=11, domain="spectral", mask_method=None, probmatching_method=None, vel_pert_method=None,
                                     num_workers=1)
    finally:
        steps_mod.nowcast_main_loop = orig
    assert len(seen) == 6 and max(seen) == 0.0, seen
