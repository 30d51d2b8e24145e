"""Callers either side of the path: extrapolation nowcast + dB transform on device.

Mirrors pysteps/tests/test_nowcasts_lagrangian_probability.py (zero-velocity identity through
nowcasts.extrapolation.forecast, ndarray / float-list timesteps) and checks the device dB
transform against the reference formula (transformation.py:150-232).
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_forecast_zero_velocity_identity_and_checks():
    from pysteps_amd.nowcasts.extrapolation import forecast

    rng = np.random.default_rng(0)
    precip = rng.random((20, 20))
    out = forecast(precip, np.zeros((2, 20, 20)), np.array([1.0, 2.0, 5.0, 12.0]))
    assert out.shape == (4, 20, 20)
    np.testing.assert_allclose(out, np.broadcast_to(precip.astype(np.float32), out.shape), rtol=0, atol=0)
    out, secs = forecast(precip, np.zeros((2, 20, 20)), [1.0, 2.0], measure_time=True)
    assert out.shape == (2, 20, 20) and secs >= 0
    holes = precip.copy()
    holes[3, 4] = np.nan  # forecast sets allow_nonfinite_values itself (:76)
    assert np.isnan(forecast(holes, np.zeros((2, 20, 20)), 1)[0, 3, 4])
    with pytest.raises(ValueError):
        forecast(precip[0], np.zeros((2, 20, 20)), 1)
    with pytest.raises(ValueError):
        forecast(precip, np.zeros((20, 20)), 1)
    with pytest.raises(ValueError):
        forecast(precip, np.zeros((2, 21, 20)), 1)
    with pytest.raises(ValueError):
        forecast(precip, np.zeros((2, 20, 20)), [2, 1])


def test_resident_chain_rate_to_db_lk_nowcast_to_rate():
    """rain rate -> dB -> dense LK -> extrapolation nowcast -> rain rate without leaving HBM,
    against the same chain through the host entry points."""
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.motion import get_method
    from pysteps_amd.nowcasts.extrapolation import forecast
    from pysteps_amd.utils import dB_transform
    from tools import synth

    m, n = 256, 256
    db = synth.rain_field_db(m, n, seed=6, sigma=3.0)
    rate0 = np.where(db > -15, 10.0 ** (db / 10.0), 0.0).astype(np.float32)
    rate1 = np.roll(rate0, (1, 2), axis=(0, 1))
    rate = np.stack([rate0, rate1])

    # host reference chain (NumPy dB transform = the reference formula)
    hdb, meta = dB_transform(rate, threshold=0.1, zerovalue=-15.0)
    want_db = np.where(rate < 0.1, -15.0, 10.0 * np.log10(np.maximum(rate, 1e-30)))
    np.testing.assert_allclose(hdb, want_db, rtol=1e-6)
    assert meta["transform"] == "dB" and meta["zerovalue"] == -15.0
    hV = get_method("LK")(hdb)
    hfc = forecast(hdb[-1], hV, 3, extrap_kwargs={"outval": "min"})
    hrate, meta2 = dB_transform(hfc, meta, inverse=True)
    assert meta2["transform"] is None

    # resident chain
    ddb, dmeta = dB_transform(DeviceArray.from_host(rate), threshold=0.1, zerovalue=-15.0)
    assert isinstance(ddb, DeviceArray) and dmeta == meta
    np.testing.assert_allclose(ddb.to_host(), hdb, rtol=2e-6, atol=2e-5)
    dV = get_method("LK")(ddb)
    dfc = forecast(ddb.view(1), dV, 3, extrap_kwargs={"outval": "min"})
    assert isinstance(dfc, DeviceArray) and dfc.shape == (3, m, n)
    drate, _ = dB_transform(dfc, dmeta, inverse=True)
    got = drate.to_host()
    assert np.isfinite(got).all()
    assert np.max(np.abs(got - hrate)) < 1e-2 * max(1.0, float(hrate.max()))
    # the motion that moved the rain is found: (2, 1) px per step
    inner = (slice(64, 192), slice(64, 192))
    assert abs(float(np.mean(dV.to_host()[0][inner])) - 2.0) < 0.1
    assert abs(float(np.mean(dV.to_host()[1][inner])) - 1.0) < 0.1


def test_field_stats_and_db_edge_cases():
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.utils import dB_transform
    from pysteps_amd.utils.transformation import field_stats

    a = np.array([[0.05, 0.1, 1.0, np.nan, 100.0, np.inf, 0.0]], dtype=np.float32)
    mn, mx, bad = field_stats(DeviceArray.from_host(a))
    assert mn == 0.0 and mx == 100.0 and bad == 2
    out, meta = dB_transform(DeviceArray.from_host(a))
    got = out.to_host()
    thr_db = 10 * np.log10(0.1)
    want = np.array([[thr_db - 5, -10.0, 0.0, np.nan, 20.0, np.inf, thr_db - 5]])
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-5)
    back, meta2 = dB_transform(out, meta, inverse=True)
    b = back.to_host()
    assert b[0, 0] == 0.0 and b[0, 6] == 0.0 and np.isnan(b[0, 3])
    np.testing.assert_allclose(b[0, [1, 2, 4]], [0.1, 1.0, 100.0], rtol=1e-5)
    same, m3 = dB_transform(out, meta)  # already in dB: unchanged
    assert same is out and m3["transform"] == "dB"
