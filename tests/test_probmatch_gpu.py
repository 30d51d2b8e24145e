"""Device probability matching (csrc/probmatch.hip, psh_probmatch_dev) against the reference's outputs
(tests/golden/probmatch_reference.npz), the oracle (oracle/probmatch.py) and the reference itself
(pysteps/postprocessing/probmatching.py:55-140, from oracle/_ref).  Everything here is selection and
copying of float64 values, so the bar is bit-exact; the single freedom is the order of TIED wet values
of the initial array, where the device is compared with the oracle's stable order."""

import os

import numpy as np
import pytest

from oracle import probmatch as oracle

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "probmatch_reference.npz")


def _match(initial, target, **kw):
    from pysteps_amd.postprocessing.probmatching import nonparam_match_empirical_cdf

    return nonparam_match_empirical_cdf(initial, target, **kw)


def _forecast_like(shape, seed, wet_fraction):
    """Continuous field with its dry part at -15 (what the member loops hand in) and a quantised one."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(seed)
    g = gaussian_filter(rng.standard_normal(shape), 6.0, mode="wrap")
    g = 12.0 * g / g.std() + rng.normal(0, 1e-2, shape)
    out = g.copy()
    out[g < np.quantile(g, 1.0 - wet_fraction)] = -15.0
    return out


def test_reference_goldens_bit_exact():
    blob = np.load(GOLDEN)
    for name in sorted({k.split("/")[0] for k in blob.files}):
        got = _match(blob[name + "/initial"], blob[name + "/target"])
        assert got.dtype == np.float64 and got.shape == blob[name + "/out"].shape
        assert np.array_equal(got, blob[name + "/out"]), name


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (5, 3), (64, 65), (300, 200), (1024, 1024)])
def test_random_fields_against_the_oracle(ref_pysteps, shape):
    from pysteps_amd.device import DeviceArray

    rng = np.random.default_rng(shape[0] * 31 + shape[1])
    for it in range(6):
        initial = rng.normal(size=shape) * 5
        if it % 3:
            initial[initial < rng.uniform(-6, 6)] = -15.0
        target = np.round(rng.normal(size=shape) * 6, int(rng.integers(0, 3)))  # quantised: heavy ties
        if it % 2:
            target[target < rng.uniform(-6, 6)] = -15.0
        if it == 4:
            target[rng.random(shape) < 0.1] = np.nan
        if it == 5:
            initial = np.round(initial, 1)  # tied wet values in the initial array: stable order
        with np.errstate(all="ignore"):  # a tiny target can be all NaN: the reference's (NaN) answer
            want = oracle.nonparam_match_empirical_cdf(initial, target)
        assert np.array_equal(_match(initial, target), want, equal_nan=True), it
        if np.isfinite(target).any():  # resident arrays always take the kernels, whatever their size
            got = _match(DeviceArray.from_host(initial), DeviceArray.from_host(target))
            assert np.array_equal(got.to_host(), want), it


def test_against_the_live_reference_and_input_dtypes(ref_pysteps):
    from pysteps.postprocessing.probmatching import nonparam_match_empirical_cdf as ref

    initial = _forecast_like((512, 384), 1, 0.3)
    target = np.round(_forecast_like((512, 384), 2, 0.45), 1)
    assert np.array_equal(_match(initial, target), ref(initial, target))          # wet area adjusted (:107-110)
    f32 = initial.astype(np.float32)
    assert np.array_equal(_match(f32, target.astype(np.float32)), ref(f32, target.astype(np.float32)))
    ints = np.arange(24, dtype=np.int64).reshape(4, 6)[::-1]
    assert np.array_equal(_match(ints, np.linspace(0, 1, 24).reshape(6, 4)), ref(ints, np.linspace(0, 1, 24).reshape(6, 4)))
    assert np.array_equal(_match(initial[::2, ::3], target[::2, ::3]), ref(initial[::2, ::3], target[::2, ::3]))


def _resident(initial, target):
    from pysteps_amd.device import DeviceArray

    return _match(DeviceArray.from_host(initial), DeviceArray.from_host(target)).to_host()


def test_degenerate_fields():
    flat = np.full((40, 50), 3.0)
    ramp = np.arange(2000.0).reshape(40, 50)
    assert np.array_equal(_match(flat, ramp), np.full((40, 50), 0.0))     # nothing wet: all at the target's zero
    assert np.array_equal(_match(ramp, flat), flat)                       # nothing wet in the target
    assert np.array_equal(_match(ramp, ramp[::-1].copy()), ramp)          # a permutation is undone
    assert np.array_equal(_match(ramp, ramp), oracle.nonparam_match_empirical_cdf(ramp, ramp))
    assert np.array_equal(_resident(flat, ramp), np.full((40, 50), 0.0))  # the same through the kernels
    assert np.array_equal(_resident(ramp, flat), flat)
    assert np.array_equal(_resident(ramp, ramp[::-1].copy()), ramp)


def test_crowded_buckets_take_the_workgroup_path():
    """One far outlier squeezes the other wet values into a few hundred of the 2^20 value buckets
    (several hundred to a few thousand values each, all different)."""
    rng = np.random.default_rng(8)
    initial = rng.random((500, 400))
    initial[17, 23] = 4000.0
    target = rng.random((500, 400))
    target[3, 5] = 2500.0
    assert np.array_equal(_match(initial, target), oracle.nonparam_match_empirical_cdf(initial, target))
    # a few distinct values in the target: single-valued buckets of any size
    coarse = np.round(rng.normal(size=(500, 400)) * 3, 0)
    assert np.array_equal(_match(initial, coarse), oracle.nonparam_match_empirical_cdf(initial, coarse))


def test_errors_and_declined_inputs(ref_pysteps):
    from pysteps.postprocessing.probmatching import nonparam_match_empirical_cdf as ref

    from pysteps_amd.device import DeviceArray

    ok = np.arange(80.0 * 90).reshape(80, 90)
    with pytest.raises(ValueError, match="Initial array contains only nans"):
        _match(np.full((80, 90), np.nan), ok)
    with pytest.raises(ValueError, match="Initial array contains only nans"):
        _match(DeviceArray.from_host(np.full((5, 6), np.nan)), DeviceArray.from_host(ok[:5, :6].copy()))
    bad = ok.copy()
    bad[2, 2] = np.nan
    with pytest.raises(ValueError, match="non-finite values outside ignore_indices"):
        _match(bad, ok)
    bad[2, 2] = np.inf
    with pytest.raises(ValueError, match="non-finite values outside ignore_indices"):
        _match(bad, ok)
    with pytest.raises(ValueError, match="dimension mismatch"):
        _match(ok, ok[:3])
    # more than 16384 tied wet values in the initial array: the device declines, the reference answers
    rng = np.random.default_rng(2)
    tied = np.where(rng.random((300, 300)) < 0.5, 1.0, -15.0) + 0.0
    tied[0, :10] = np.arange(10) + 2.0
    target = rng.normal(size=(300, 300))
    assert np.array_equal(_match(tied, target), ref(tied, target))
    # resident arrays: the one declined call crosses the bus, runs the reference's function and comes back
    # (a resident member loop with a plateaued forecast keeps going)
    resident = _match(DeviceArray.from_host(tied), DeviceArray.from_host(target))
    assert isinstance(resident, DeviceArray) and np.array_equal(resident.to_host(), ref(tied, target))
    inf_target = target.copy()
    inf_target[4, 4] = np.inf
    initial = rng.normal(size=(300, 300))
    assert np.array_equal(_match(initial, inf_target), ref(initial, inf_target))
    ignore = rng.random((300, 300)) < 0.1
    assert np.array_equal(_match(initial, target, ignore_indices=ignore), ref(initial, target, ignore_indices=ignore))


def test_full_size_resident_and_properties():
    """BASELINE size (4096 x 4096): device-resident call, checked against the oracle and through the
    function's own invariants (ranks kept, zeros kept, values drawn from the target)."""
    from pysteps_amd.device import DeviceArray

    shape = (4096, 4096)
    initial = _forecast_like(shape, 5, 0.25)
    target = np.round(_forecast_like(shape, 6, 0.35), 1)
    d_out = _match(DeviceArray.from_host(initial), DeviceArray.from_host(target))
    assert isinstance(d_out, DeviceArray) and d_out.shape == shape
    got = d_out.to_host()
    assert np.array_equal(got, _match(initial, target))
    dry = initial == initial.min()
    assert np.all(got[dry] == target.min())
    order = np.argsort(initial[~dry], kind="stable")
    assert np.all(np.diff(got[~dry][order]) >= 0)                     # ranks of the wet pixels are kept
    top = np.sort(target, axis=None)[-(~dry).sum():]                  # they receive the top of the target,
    received = np.sort(got[~dry])                                     # its values below the threshold zeroed
    assert np.all((received == top) | (received == target.min()))
    assert np.array_equal(got, oracle.nonparam_match_empirical_cdf(initial, target))


def test_nowcasts_steps_with_the_patched_matching(ref_pysteps):
    """nowcasts.steps (probmatching_method="cdf", steps.py:1199) with the module attribute replaced
    gives the stock result: the forecasts it matches are continuous (no tied wet values)."""
    from pysteps import nowcasts

    from pysteps_amd import register
    from tools import synth

    frames = synth.steps_frames(256, 256, 3)
    V = synth.true_velocity(256, 256).astype(np.float64)
    kw = dict(n_ens_members=2, n_cascade_levels=6, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, seed=7,
              vel_pert_method=None, mask_method="incremental", probmatching_method="cdf", num_workers=1)
    steps = nowcasts.get_method("steps")
    want = steps(frames, V, 3, **kw)
    try:
        assert register.patch_probmatching()
        got = steps(frames, V, 3, **kw)
    finally:
        register.unpatch_probmatching()
    assert np.array_equal(got, want, equal_nan=True)


def test_async_variant_reports_its_outcome_in_device_memory(ref_pysteps):
    """psh_probmatch_async_dev (the resident member loop's form: no wait, the outcome in a device word)"""
    from pysteps.postprocessing.probmatching import nonparam_match_empirical_cdf as ref

    from pysteps_amd import _lib
    from pysteps_amd.device import DeviceArray

    lib = _lib.lib()
    rng = np.random.default_rng(12)
    shape = (300, 300)
    initial, target = _forecast_like(shape, 3, 0.3), np.round(_forecast_like(shape, 4, 0.4), 1)
    tied = np.where(rng.random(shape) < 0.5, 1.0, -15.0) + 0.0  # 45000 tied wet values: declined
    status = DeviceArray((4,), np.int32)
    d_t = DeviceArray.from_host(target)
    outs = []
    for k, arr in enumerate((initial, tied, np.full(shape, np.nan))):
        d_i, d_o = DeviceArray.from_host(arr), DeviceArray(shape, np.float64)
        _lib.check(lib.psh_probmatch_async_dev(d_i.ptr, d_t.ptr, arr.size, d_o.ptr, status.ptr + 4 * k))
        outs.append((d_i, d_o))
    got = status.to_host()
    assert got[0] == 0 and np.array_equal(outs[0][1].to_host(), ref(initial, target))
    assert lib.psh_probmatch_status(int(got[1])) == _lib.PSH_EUNSUPPORTED
    assert lib.psh_probmatch_status(int(got[2])) == _lib.PSH_EINVAL and "only nans" in _lib.last_error()


def test_plans_for_a_fixed_target_give_the_same_outputs(ref_pysteps):
    """psh_probmatch_plan_create / _planned_dev (the target's half of the work done once: every member of a
    STEPS ensemble is matched against the same observation at every time step) against psh_probmatch_dev."""
    import ctypes

    from pysteps.postprocessing.probmatching import nonparam_match_empirical_cdf as ref

    from pysteps_amd import _lib
    from pysteps_amd.device import DeviceArray

    lib = _lib.lib()
    rng = np.random.default_rng(21)
    shape = (300, 260)
    count = shape[0] * shape[1]
    targets = [np.round(_forecast_like(shape, 4, 0.4), 1),       # quantised observation: crowded buckets
               _forecast_like(shape, 5, 0.1),                     # wet area below most forecasts: no adjustment
               np.where(rng.random(shape) < 0.02, np.nan, _forecast_like(shape, 6, 0.5))]  # NaNs count as zeros
    for t_no, target in enumerate(targets):
        d_t = DeviceArray.from_host(target)
        plan = ctypes.c_void_p()
        _lib.check(lib.psh_probmatch_plan_create(d_t.ptr, count, ctypes.byref(plan)))
        try:
            status = DeviceArray((4,), np.int32)
            tied = np.where(rng.random(shape) < 0.5, 1.0, -15.0) + 0.0
            cases = [_forecast_like(shape, 30 + t_no, 0.3), _forecast_like(shape, 40 + t_no, 0.7), tied, np.full(shape, np.nan)]
            kept = []
            for k, initial in enumerate(cases):
                d_i, d_o = DeviceArray.from_host(initial), DeviceArray(shape, np.float64)
                _lib.check(lib.psh_probmatch_planned_dev(plan, d_i.ptr, count, d_o.ptr, status.ptr + 4 * k))
                kept.append((d_i, d_o))
            got = status.to_host()
            for k in (0, 1):
                assert got[k] == 0
                with np.errstate(all="ignore"):
                    np.testing.assert_array_equal(kept[k][1].to_host(), ref(cases[k], target))
                direct = DeviceArray(shape, np.float64)
                _lib.check(lib.psh_probmatch_dev(kept[k][0].ptr, d_t.ptr, count, direct.ptr))
                np.testing.assert_array_equal(kept[k][1].to_host(), direct.to_host())
            assert lib.psh_probmatch_status(int(got[2])) == _lib.PSH_EUNSUPPORTED
            assert lib.psh_probmatch_status(int(got[3])) == _lib.PSH_EINVAL and "only nans" in _lib.last_error()
            # the waiting form returns the verdict itself
            d_o = DeviceArray(shape, np.float64)
            assert lib.psh_probmatch_planned_dev(plan, kept[0][0].ptr, count, d_o.ptr, None) == 0
            np.testing.assert_array_equal(d_o.to_host(), kept[0][1].to_host())
            assert lib.psh_probmatch_planned_dev(plan, kept[2][0].ptr, count, d_o.ptr, None) == _lib.PSH_EUNSUPPORTED
            assert lib.psh_probmatch_planned_dev(plan, kept[0][0].ptr, count - 1, d_o.ptr, None) == _lib.PSH_EINVAL
        finally:
            _lib.check(lib.psh_probmatch_plan_destroy(plan))
    # a target the kernels decline (infinity): the plan carries the verdict to every call
    bad = _forecast_like(shape, 8, 0.3)
    bad[3, 3] = np.inf
    d_t = DeviceArray.from_host(bad)
    plan = ctypes.c_void_p()
    _lib.check(lib.psh_probmatch_plan_create(d_t.ptr, count, ctypes.byref(plan)))
    d_i, d_o = DeviceArray.from_host(_forecast_like(shape, 9, 0.3)), DeviceArray(shape, np.float64)
    assert lib.psh_probmatch_planned_dev(plan, d_i.ptr, count, d_o.ptr, None) == _lib.PSH_EUNSUPPORTED
    _lib.check(lib.psh_probmatch_plan_destroy(plan))
