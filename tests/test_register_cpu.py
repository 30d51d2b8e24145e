"""Registration into pysteps' method tables (no GPU: nothing is computed)."""

import pytest

from pysteps_amd import register
from pysteps_amd.extrapolation.semilagrangian import extrapolate
from pysteps_amd.motion.lucaskanade import dense_lucaskanade
from tools import ref_loader


def test_register_into_plain_dicts():
    motion, extrap = {"lk": object()}, {"semilagrangian": object()}
    stock_lk, stock_sl = motion["lk"], extrap["semilagrangian"]
    added = register.register_into(motion, extrap)
    assert motion["lk_hip"] is dense_lucaskanade and motion["lucaskanade_hip"] is dense_lucaskanade
    assert extrap["semilagrangian_hip"] is extrapolate
    assert motion["lk"] is stock_lk and extrap["semilagrangian"] is stock_sl  # default: non-overriding
    assert "motion:lk_hip" in added and "extrapolation:semilagrangian_hip" in added
    register.register_into(motion, extrap, override=True)
    assert motion["lk"] is dense_lucaskanade and motion["lucaskanade"] is dense_lucaskanade
    assert extrap["semilagrangian"] is extrapolate


@pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")
def test_reference_get_method_serves_the_hip_extrapolator():
    """The real pysteps.extrapolation.interface.get_method returns our callable by name."""
    iface = ref_loader.load("pysteps.extrapolation.interface")
    saved = dict(iface._extrapolation_methods)
    try:
        register.register_into(None, iface._extrapolation_methods)
        assert iface.get_method("semilagrangian_hip") is extrapolate
        assert iface.get_method("SEMILAGRANGIAN_HIP") is extrapolate  # names are case-insensitive
        assert iface.get_method("semilagrangian") is saved["semilagrangian"]
        assert iface.get_method(None) is saved[None]
        with pytest.raises(ValueError):
            iface.get_method("nonexistent")
    finally:
        iface._extrapolation_methods.clear()
        iface._extrapolation_methods.update(saved)


def test_own_interfaces_mirror_the_reference_contract():
    from pysteps_amd import extrapolation, motion

    assert extrapolation.get_method("semilagrangian") is extrapolate
    assert extrapolation.get_method("SemiLagrangian") is extrapolate
    assert extrapolation.get_method(None)(None, None, 1) is None
    assert extrapolation.get_method("eulerian") is not None
    assert motion.get_method("LK") is dense_lucaskanade
    assert motion.get_method("lucaskanade") is dense_lucaskanade
    with pytest.raises(ValueError):
        extrapolation.get_method("unknown")
    with pytest.raises(ValueError):
        motion.get_method("unknown")


def test_fft_method_name_resolves_through_the_reference_lookup(ref_pysteps):
    """register() makes pysteps.utils.get_method("hip", shape=...) answer (the reference hard-codes
    its FFT method names, interface.py:240-243, so the lookup function is wrapped); shapes the
    kernels do not take are served by numpy.fft - which is all that can run without a GPU."""
    import numpy as np
    from pysteps import utils
    from pysteps.utils import interface

    try:
        added = register.register()
        assert "fft:hip" in added or hasattr(interface.get_method, "_pysteps_amd_reference")
        assert utils.get_method is interface.get_method
        fft = utils.get_method("hip", shape=(5000, 6), n_threads=4)  # a side beyond the kernels: numpy.fft serves it
        x = np.random.default_rng(0).standard_normal((5000, 6))
        assert np.array_equal(fft.rfft2(x), np.fft.rfft2(x))
        assert np.array_equal(fft.irfft2(np.fft.rfft2(x)), np.fft.irfft2(np.fft.rfft2(x), s=(5000, 6)))
        assert np.array_equal(fft.fft2(x), np.fft.fft2(x)) and np.array_equal(fft.ifft2(x), np.fft.ifft2(x))
        assert np.array_equal(fft.fftshift(x), np.fft.fftshift(x))
        with pytest.raises(KeyError):
            utils.get_method("hip")
        # everything else is the reference's own answer
        ref = utils.get_method("numpy", shape=(64, 64))
        assert ref.rfft2 is np.fft.rfft2
        assert utils.get_method("dB") is ref_pysteps.utils.transformation.dB_transform
        with pytest.raises(ValueError):
            utils.get_method("no_such_method")
    finally:
        register.unregister_fft()
    assert not hasattr(interface.get_method, "_pysteps_amd_reference")
    with pytest.raises(ValueError):
        utils.get_method("hip", shape=(64, 64))


def test_feature_detectors_resolve_through_the_reference_lookup(ref_pysteps):
    """register() adds "blob_hip" / "shitomasi_hip" to pysteps.feature's table (feature/interface.py:26-29); the stock
    names keep the reference's functions unless override is asked for."""
    import pysteps.feature.interface as feat_if
    from pysteps import feature

    from pysteps_amd.feature import blob, shitomasi

    saved = dict(feat_if._detection_methods)
    try:
        added = register.register_features()
        assert set(added) == {"feature:blob_hip", "feature:shitomasi_hip"}
        assert feature.get_method("blob_hip") is blob.detection and feature.get_method("Shitomasi_HIP") is shitomasi.detection
        assert feature.get_method("blob") is saved["blob"] and feature.get_method("shitomasi") is saved["shitomasi"]
        register.register_features(override=True)
        assert feature.get_method("blob") is blob.detection and feature.get_method("shitomasi") is shitomasi.detection
        assert feature.get_method("tstorm") is saved["tstorm"]
    finally:
        feat_if._detection_methods.clear()
        feat_if._detection_methods.update(saved)
