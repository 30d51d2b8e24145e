"""AR(p) step mirror without a GPU: series the device path does not take have to come back from the
reference's own function (pysteps/timeseries/autoregression.py:1020-1070), with its errors; the
registration hook swaps and restores the module attribute."""

import numpy as np
import pytest


def test_small_and_odd_series_are_the_reference(ref_pysteps):
    from pysteps.timeseries.autoregression import iterate_ar_model as ref

    from pysteps_amd.timeseries.autoregression import iterate_ar_model

    rng = np.random.default_rng(1)
    x = rng.normal(size=(3, 40, 50))
    eps = rng.normal(size=(40, 50))
    phi = [0.7, -0.2, 0.05, 0.4]
    assert np.array_equal(iterate_ar_model(x, phi, eps=eps), ref(x, phi, eps=eps))          # small field
    assert np.array_equal(iterate_ar_model(x, phi), ref(x, phi))
    x1 = rng.normal(size=5)  # 1-D series: the reference's own hstack error (:1067) comes through
    for fn in (ref, iterate_ar_model):
        with pytest.raises(ValueError, match="concatenation axis"):
            fn(x1, [0.5, 0.1, 0.3])
    big32 = rng.normal(size=(2, 300, 300)).astype(np.float32)
    assert np.array_equal(iterate_ar_model(big32, [0.5, 0.1, 0.3]), ref(big32, [0.5, 0.1, 0.3]))       # not float64
    per_pixel = [rng.normal(size=(300, 300)), rng.normal(size=(300, 300))]
    big = rng.normal(size=(1, 300, 300))
    assert np.array_equal(iterate_ar_model(big, per_pixel, eps=big[0]), ref(big, per_pixel, eps=big[0]))
    with pytest.raises(ValueError, match="dimension mismatch between x and phi"):
        iterate_ar_model(x[:1], phi)
    with pytest.raises(ValueError, match="dimension mismatch between x and eps"):
        iterate_ar_model(x, phi, eps=eps[:10])


def test_patch_and_unpatch(ref_pysteps):
    import pysteps.timeseries.autoregression as ref_mod

    from pysteps_amd import register
    from pysteps_amd.timeseries import autoregression as hip_mod

    stock = ref_mod.iterate_ar_model
    try:
        assert register.patch_autoregression() == ["autoregression:iterate_ar_model"]
        assert ref_mod.iterate_ar_model is hip_mod.iterate_ar_model
        assert register.patch_autoregression() == []
        assert hip_mod._reference() is stock
    finally:
        register.unpatch_autoregression()
    assert ref_mod.iterate_ar_model is stock and not hasattr(ref_mod, "_reference_iterate_ar_model")
