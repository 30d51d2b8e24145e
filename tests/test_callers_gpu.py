"""The REAL pysteps callers driving the HIP operators by method name.

``oracle/_ref`` is the unmodified reference package (``oracle/build_ref.py``); after
``pysteps_amd.register.register()`` the callers named by ``north_star`` pick the HIP extrapolator
up through their ``extrap_method=`` string:

* ``pysteps.nowcasts.extrapolation.forecast``        (nowcasts/extrapolation.py:90-92)
* ``pysteps.nowcasts.steps.forecast``                (nowcasts/steps.py:656,697-720 + the generic loop)
* ``pysteps.nowcasts.utils.nowcast_main_loop``       (nowcasts/utils.py:441-462, 489-503) via S-PROG too
* ``pysteps.nowcasts.lagrangian_probability``        (ndarray timesteps, :67-68)

Every result is compared with the SAME caller running the stock ``"semilagrangian"``.
Tolerance: 1e-4 relative L2 (BASELINE.json), identical NaN masks.
"""

import numpy as np
import pytest

from conftest import nan_mismatch, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pysteps(ref_pysteps):
    from pysteps_amd import register

    added = register.register()  # non-overriding: adds "semilagrangian_hip", "lk_hip", "lucaskanade_hip"
    assert "extrapolation:semilagrangian_hip" in added
    return ref_pysteps


def test_registered_names_resolve_through_the_real_interfaces(pysteps):
    from pysteps import extrapolation, motion
    from pysteps_amd.extrapolation.semilagrangian import extrapolate
    from pysteps_amd.motion.lucaskanade import dense_lucaskanade

    assert extrapolation.get_method("semilagrangian_hip") is extrapolate
    assert extrapolation.get_method("SemiLagrangian_HIP") is extrapolate  # lower-cased by the reference
    assert motion.get_method("LK_hip") is dense_lucaskanade
    # the stock names are untouched (pysteps/tests/test_interfaces.py:69-78 keeps passing)
    from pysteps.extrapolation.semilagrangian import extrapolate as stock

    assert extrapolation.get_method("semilagrangian") is stock


@pytest.mark.parametrize("variant", ["config1", "nan_field", "n_iter3_list"])
def test_nowcasts_extrapolation_forecast(pysteps, variant):
    """BASELINE config 1 (512 x 512, 6 lead times) through nowcasts.get_method("extrapolation")."""
    from pysteps import nowcasts
    from tools import synth

    m = n = 512
    P = synth.rain_field_db(m, n, seed=11)
    V = synth.true_velocity(m, n)
    timesteps, kw = 6, {}
    if variant == "nan_field":
        P = P.copy()
        P[synth.border_nan_mask(m, n, 0.1)] = np.nan  # forecast() sets allow_nonfinite_values itself (:76)
    elif variant == "n_iter3_list":
        timesteps, kw = [0.5, 1.0, 2.5, 4.0], {"n_iter": 3, "outval": -15.0}
    fc = nowcasts.get_method("extrapolation")
    want = fc(P, V, timesteps, extrap_method="semilagrangian", extrap_kwargs=dict(kw))
    got = fc(P, V, timesteps, extrap_method="semilagrangian_hip", extrap_kwargs=dict(kw))
    assert got.shape == want.shape and got.dtype == want.dtype
    assert nan_mismatch(got, want) == 0
    assert rel_l2(got, want) < 1e-4


def _steps_inputs(m, n):
    from tools import synth

    frames = synth.steps_frames(m, n, 3)
    V = synth.true_velocity(m, n).astype(np.float64)
    return frames, V


def _steps_kwargs():
    return dict(n_ens_members=4, n_cascade_levels=6, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0,
                seed=42, vel_pert_method="bps", mask_method="incremental", num_workers=1)


FLIPS_SEEN = []  # (test, NaN-mask mismatches / pixels, pixels off by > 1 % of the range / pixels)


def _ensemble_close(got, want, field_tol=1e-4, flip_frac=2e-6):
    """STEPS thresholds (precip mask, incremental mask) and rank-matches (CDF matching) its fields:
    a 1e-7 difference in an advected value can move a pixel across such a decision.  Compare the
    pixels that took the same side within 1e-4 rel-L2 and bound the fraction that did not: observed
    on MI355X: none at all (0 NaN-mask mismatches, 0 pixels off by more than 1 % of the range in every
    caller test of this file, profiles/r04/x_flips_seen.json), bound 2e-6 of the pixels;
    the observed fractions are collected in FLIPS_SEEN and printed when a bound fails."""
    import inspect

    assert got.shape == want.shape
    nan_frac = nan_mismatch(got, want) / want.size
    diff = np.abs(got - want)
    ok = np.isfinite(diff)
    scale = float(np.nanmax(want) - np.nanmin(want))
    flipped = ok & (diff > 1e-2 * scale)
    frac = np.count_nonzero(flipped) / want.size
    FLIPS_SEEN.append((inspect.stack()[1].function, nan_frac, frac))
    assert nan_frac <= flip_frac, FLIPS_SEEN
    assert frac <= flip_frac, FLIPS_SEEN
    same = ok & ~flipped
    return float(np.linalg.norm((got - want)[same]) / np.linalg.norm(want[same]))


def test_nowcasts_steps_small_ensemble(pysteps):
    """nowcasts.steps: 256 x 256, 4 members, seed 42, BPS velocity perturbations - the generic loop
    calls the extrapolator with (precip_j, V_j, [dt], xy_coords, displacement_prev, ...) per member
    and time step (nowcasts/utils.py:453-458) and steps.py:697-703 calls it positionally with
    outval="min"."""
    from pysteps import nowcasts

    frames, V = _steps_inputs(256, 256)
    steps = nowcasts.get_method("steps")
    want = steps(frames, V, 3, extrap_method="semilagrangian", **_steps_kwargs())
    got = steps(frames, V, 3, extrap_method="semilagrangian_hip", **_steps_kwargs())
    assert want.shape == (4, 3, 256, 256)
    rel = _ensemble_close(got, want)
    assert rel < 1e-4, rel


def test_nowcasts_steps_subtimesteps_and_no_vel_pert(pysteps):
    """List timesteps make the loop advect by fractional increments and use the displacement-only
    call (precip None, nowcasts/utils.py:489-503)."""
    from pysteps import nowcasts

    frames, V = _steps_inputs(192, 256)
    steps = nowcasts.get_method("steps")
    kw = _steps_kwargs()
    kw.update(vel_pert_method=None, n_ens_members=2)
    ts = [0.5, 1.0, 2.5]
    want = steps(frames, V, ts, extrap_method="semilagrangian", **kw)
    got = steps(frames, V, ts, extrap_method="semilagrangian_hip", **kw)
    rel = _ensemble_close(got, want)
    assert rel < 1e-4, rel


def test_nowcast_main_loop_through_sprog(pysteps):
    """S-PROG drives the same nowcast_main_loop deterministically (sprog.py:204)."""
    from pysteps import nowcasts

    frames, V = _steps_inputs(256, 256)
    sprog = nowcasts.get_method("sprog")
    kw = dict(n_cascade_levels=6, precip_thr=-10.0)
    want = sprog(frames, V, 4, extrap_method="semilagrangian", **kw)
    got = sprog(frames, V, 4, extrap_method="semilagrangian_hip", **kw)
    rel = _ensemble_close(got, want)
    assert rel < 1e-4, rel


def test_lagrangian_probability_ndarray_timesteps(pysteps):
    from pysteps.nowcasts import lagrangian_probability as lp
    from tools import synth

    m, n = 128, 160
    P = np.where(synth.rain_field_db(m, n, seed=5, sigma=3.0) > -15, 5.0, 0.0)
    V = synth.true_velocity(m, n).astype(np.float64)
    want = lp.forecast(P, V, 3, threshold=1.0, slope=1.0)
    # the module calls extrapolation.get_method("semilagrangian"): exercise override=True
    from pysteps import extrapolation
    from pysteps_amd import register

    table = extrapolation.interface._extrapolation_methods
    stock = table["semilagrangian"]
    try:
        register.register(override=True)
        assert extrapolation.get_method("semilagrangian") is not stock
        got = lp.forecast(P, V, 3, threshold=1.0, slope=1.0)
    finally:
        table["semilagrangian"] = stock
        from pysteps import motion

        from pysteps.motion.lucaskanade import dense_lucaskanade as stock_lk

        motion.interface._methods["lk"] = motion.interface._methods["lucaskanade"] = stock_lk
    assert nan_mismatch(got, want) == 0
    assert rel_l2(got, want) < 1e-4


def test_motion_get_method_lk_hip_on_steps_frames(pysteps):
    """motion.get_method("lk_hip") through the real interface; cv2 is absent, so the stock "LK"
    cannot run beside it - the OpenCV restatement is the checker (parity unpinned)."""
    from oracle import lk_opencv as olk
    from pysteps import motion

    frames, V = _steps_inputs(256, 256)
    got = motion.get_method("LK_hip")(frames)
    want = olk.dense_lucaskanade(frames)
    assert got.shape == (2, 256, 256) and got.dtype == np.float64
    assert rel_l2(got, want) < 1e-3


def test_db_transform_and_forecast_mirrors_match_the_reference(pysteps):
    """pysteps_amd's own dB_transform / nowcasts.extrapolation.forecast mirrors
    against the reference's functions, host and device containers."""
    from pysteps.nowcasts.extrapolation import forecast as ref_forecast
    from pysteps.utils.transformation import dB_transform as ref_db
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.nowcasts.extrapolation import forecast
    from pysteps_amd.utils import dB_transform
    from tools import synth

    m, n = 200, 264
    db = synth.rain_field_db(m, n, seed=8, sigma=3.0)
    rate = np.where(db > -15, 10.0 ** (db / 10.0), 0.0).astype(np.float32)
    rate[3, 5] = np.nan
    meta0 = {"transform": None, "threshold": 0.1, "zerovalue": 0.0, "unit": "mm/h"}
    for kw in (dict(), dict(threshold=0.5), dict(threshold=0.1, zerovalue=-15.0)):
        want, wmeta = ref_db(rate.copy(), dict(meta0), **kw)
        got, gmeta = dB_transform(rate.copy(), dict(meta0), **kw)
        assert np.array_equal(got, want, equal_nan=True) and gmeta == wmeta
        dev, dmeta = dB_transform(DeviceArray.from_host(rate), dict(meta0), **kw)
        assert dmeta == wmeta
        np.testing.assert_allclose(dev.to_host(), want, rtol=2e-6, atol=2e-5)
        back_w, bw = ref_db(want.copy(), dict(wmeta), inverse=True)
        back_g, bg = dB_transform(got.copy(), dict(gmeta), inverse=True)
        assert np.array_equal(back_g, back_w, equal_nan=True) and bg == bw
        back_d, bd = dB_transform(dev, dict(dmeta), inverse=True)
        assert bd == bw
        np.testing.assert_allclose(back_d.to_host(), back_w, rtol=1e-5, atol=1e-6)
    V = synth.true_velocity(m, n)
    dbn, _ = ref_db(rate.copy(), dict(meta0), threshold=0.1, zerovalue=-15.0)
    want = ref_forecast(dbn, V, 4, extrap_kwargs={"outval": -15.0})
    got = forecast(dbn, V, 4, extrap_kwargs={"outval": -15.0})
    assert nan_mismatch(got, want) == 0 and rel_l2(got, want) < 1e-4
    dgot = forecast(DeviceArray.from_host(dbn), DeviceArray.from_host(V), 4, extrap_kwargs={"outval": -15.0})
    assert nan_mismatch(dgot.to_host(), want) == 0 and rel_l2(dgot.to_host(), want) < 1e-4


def test_ensemble_advector_against_reference_bps_and_worker(pysteps):
    """EnsembleAdvector with perturbators from the REAL initialize_bps against the reference worker
    recipe (nowcasts/utils.py:441-462): V_j = V + generate_bps(p_j, t); extrapolate(precip_j, V_j,
    [dt], displacement_prev=D_j)."""
    from pysteps.extrapolation.semilagrangian import extrapolate as ref_extrapolate
    from pysteps.noise.motion import generate_bps, initialize_bps
    from pysteps_amd.extrapolation.ensemble import EnsembleAdvector
    from tools import synth

    B, m, n = 4, 96, 160
    members = np.stack([synth.rain_field_db(m, n, seed=70 + j, sigma=2.0) for j in range(B)])
    V = synth.true_velocity(m, n)
    V[:, 9, 11] = 0.0
    V64 = V.astype(np.float64)
    perts = []
    for j in range(B):
        rs = np.random.RandomState(1000 + j)
        perts.append(initialize_bps(V64, 1.0, 5.0, randstate=rs))  # steps.py:907-915 usage
    adv = EnsembleAdvector(V, B, perts, n_iter=1)
    D = [None] * B
    for dt, t_total in [(1.0, 5.0), (1.0, 10.0), (0.5, 12.5)]:
        got = adv.step(members, dt, t_total)
        for j in range(B):
            Vj = V64 + generate_bps(perts[j], t_total)
            want, D[j] = ref_extrapolate(members[j], Vj, [dt], return_displacement=True, displacement_prev=D[j])
            assert nan_mismatch(got[j], want[0]) <= 2
            assert rel_l2(got[j], want[0]) < 1e-4
    gd = adv.displacement.to_host()
    for j in range(B):
        assert np.max(np.abs(gd[j] - D[j])) < 1e-4


@pytest.mark.parametrize("timesteps,vel_pert", [(3, "bps"), ([0.5, 1.0, 2.5, 4.75], "bps"), (3, None)])
def test_steps_with_the_resident_member_batched_loop(pysteps, timesteps, vel_pert):
    """register(patch_main_loop=True): nowcasts.steps drives ONE member-batched launch per time step
    (pysteps_amd.nowcasts.utils -> EnsembleAdvector), perturbators taken from the real
    initialize_bps closures, trajectories resident in HBM - against the stock loop with the stock
    extrapolator (nowcasts/utils.py:441-503, noise/motion.py:146-180)."""
    from pysteps import nowcasts
    from pysteps.nowcasts import steps as steps_mod
    from pysteps_amd import register
    from pysteps_amd.extrapolation import ensemble
    from pysteps_amd.nowcasts.utils import nowcast_main_loop

    frames, V = _steps_inputs(256, 256)
    kw = _steps_kwargs()
    kw["vel_pert_method"] = vel_pert
    steps = nowcasts.get_method("steps")
    want = steps(frames, V, timesteps, extrap_method="semilagrangian", **kw)
    launches = []
    orig_step = ensemble.EnsembleAdvector.step

    def counting_step(self, *a, **k):
        launches.append(self.n_members)
        return orig_step(self, *a, **k)

    try:
        register.register(patch_main_loop=True)
        assert steps_mod.nowcast_main_loop is nowcast_main_loop
        ensemble.EnsembleAdvector.step = counting_step
        got = steps(frames, V, timesteps, extrap_method="semilagrangian_hip", **kw)
    finally:
        ensemble.EnsembleAdvector.step = orig_step
        register.unpatch_main_loop()
    assert launches and all(b == kw["n_ens_members"] for b in launches)  # the batched route ran
    assert got.dtype == want.dtype
    rel = _ensemble_close(got, want)
    assert rel < 1e-4, rel


def test_steps_with_every_device_piece_at_once(pysteps):
    """nowcasts.steps with all the operators this library offers for its member loop switched on
    together - extrapolator and resident member-batched loop, FFT method, cascade decomposition, noise
    generator by name, AR(p) step, CDF matching and incremental mask through the patched module attributes - against the
    stock run with the same seed (steps.py:637-720, 1095-1199)."""
    from pysteps import nowcasts
    from pysteps_amd import register

    frames, V = _steps_inputs(256, 256)
    kw = _steps_kwargs()
    steps = nowcasts.get_method("steps")
    want = steps(frames, V, 3, extrap_method="semilagrangian", **kw)
    try:
        added = register.register(patch_main_loop=True, probmatching=True, autoregression=True, dilated_mask=True)
        assert "probmatching:nonparam_match_empirical_cdf" in added and "autoregression:iterate_ar_model" in added
        assert "nowcasts.utils:compute_dilated_mask" in added
        got = steps(frames, V, 3, extrap_method="semilagrangian_hip", fft_method="hip", decomp_method="fft_hip",
                    noise_method="nonparametric_hip", **kw)
    finally:
        register.unpatch_main_loop()
        register.unpatch_probmatching()
        register.unpatch_autoregression()
        register.unpatch_dilated_mask()
    assert got.dtype == want.dtype
    rel = _ensemble_close(got, want)
    assert rel < 1e-4, rel


def test_check_norain_mirror_matches_the_reference(pysteps):
    """utils/check_norain.py:6-58 against pysteps_amd.utils.check_norain, host and device."""
    from pysteps.utils.check_norain import check_norain as ref_check
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.utils import check_norain
    from tools import synth

    m, n = 180, 200
    db = synth.rain_field_db(m, n, seed=12, sigma=3.0)
    dry = np.full((m, n), -15.0, np.float32)
    holes = db.copy()
    holes[:20] = np.nan
    cases = [(db, None, 0.0), (db, -10.0, 0.0), (db, 5.0, 0.3), (db, 5.0, 0.01), (dry, None, 0.0), (dry, -15.0, 0.0),
             (holes, None, 0.0), (holes, 0.0, 0.25)]
    for arr, thr, frac in cases:
        want = ref_check(arr, thr, frac, None, False)
        assert check_norain(arr, thr, frac, None, False) == want
        assert check_norain(DeviceArray.from_host(arr), thr, frac, None, False) == want
    for win in ("hann", "tukey"):
        assert check_norain(db, -10.0, 0.0, win, False) == ref_check(db, -10.0, 0.0, win, False)
        assert check_norain(DeviceArray.from_host(db), -10.0, 0.2, win, False) == ref_check(db, -10.0, 0.2, win, False)
    stack = np.stack([db, dry])
    assert check_norain(stack, -10.0, 0.0, None, False) == ref_check(stack, -10.0, 0.0, None, False)


def test_zz_flip_fractions_observed_in_this_file():
    """what _ensemble_close saw over the caller tests above (kept in gpurun_out/ for DESIGN.md)"""
    import json
    import os

    from conftest import ROOT

    assert FLIPS_SEEN
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "flips_seen.json"), "w") as fh:
        json.dump(FLIPS_SEEN, fh)
    assert max(f[1] for f in FLIPS_SEEN) <= 2e-5 and max(f[2] for f in FLIPS_SEEN) <= 2e-5, FLIPS_SEEN
