"""Device AR(p) step (csrc/cascade.hip, psh_ar_iterate_dev) against the reference's function
(pysteps/timeseries/autoregression.py:1020-1070) from oracle/_ref: float64 multiply-then-add in the
reference's order, so the bar is bit-exact."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("order,nt,shape,with_eps", [(1, 1, (300, 300), True), (2, 2, (256, 256), True),
                                                     (2, 3, (512, 384), False), (3, 3, (1024, 1024), True),
                                                     (8, 9, (260, 270), True), (2, 2, (4, 128, 256), True)])
def test_bit_exact_with_the_reference(ref_pysteps, order, nt, shape, with_eps):
    from pysteps.timeseries.autoregression import iterate_ar_model as ref

    from pysteps_amd.device import DeviceArray
    from pysteps_amd.timeseries.autoregression import iterate_ar_model

    rng = np.random.default_rng(order * 100 + nt)
    x = rng.normal(size=(nt,) + shape) * 3
    x[0].flat[::7] = -0.0
    x[-1].flat[::5] = 0.0
    phi = np.append(rng.uniform(-0.9, 0.9, order), 0.43)
    eps = rng.normal(size=shape) if with_eps else None
    want = ref(x, phi, eps=eps)
    got = iterate_ar_model(x, phi, eps=eps)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want) and np.array_equal(np.signbit(got), np.signbit(want))
    assert np.array_equal(iterate_ar_model(x, list(phi), eps=eps), want)
    d = iterate_ar_model(DeviceArray.from_host(x), phi, eps=None if eps is None else DeviceArray.from_host(eps))
    assert isinstance(d, DeviceArray) and np.array_equal(d.to_host(), want)


def test_errors_and_resident_limits(ref_pysteps):
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.timeseries.autoregression import iterate_ar_model

    x = np.random.default_rng(0).normal(size=(2, 300, 300))
    with pytest.raises(ValueError, match="dimension mismatch between x and phi"):
        iterate_ar_model(x, [0.5, 0.2, 0.1, 0.3])
    with pytest.raises(ValueError, match="dimension mismatch between x and eps"):
        iterate_ar_model(x, [0.5, 0.2, 0.3], eps=np.zeros((300, 299)))
    with pytest.raises(NotImplementedError):
        iterate_ar_model(DeviceArray.from_host(x), [0.5, 0.2, 0.3], eps=np.zeros((300, 300)))


def test_nowcasts_steps_with_the_patched_member_loop_pieces(ref_pysteps):
    """nowcasts.steps with the AR(p) step, the CDF matching and the incremental mask replaced by the
    device versions (steps.py:1095,1137,1199,1210): the stock result, bit for bit."""
    from pysteps import nowcasts

    from pysteps_amd import register
    from tools import synth

    frames = synth.steps_frames(256, 256, 3)
    V = synth.true_velocity(256, 256).astype(np.float64)
    kw = dict(n_ens_members=2, n_cascade_levels=6, precip_thr=-10.0, kmperpixel=1.0, timestep=5.0, seed=11,
              vel_pert_method="bps", mask_method="incremental", probmatching_method="cdf", num_workers=1)
    steps = nowcasts.get_method("steps")
    want = steps(frames, V, 3, **kw)
    try:
        assert register.patch_autoregression() and register.patch_probmatching() and register.patch_dilated_mask()
        got = steps(frames, V, 3, **kw)
    finally:
        register.unpatch_autoregression()
        register.unpatch_probmatching()
        register.unpatch_dilated_mask()
    assert np.array_equal(got, want, equal_nan=True)
