"""Plug the HIP operators into pysteps' own method tables.

pysteps looks the two hot-path operators up by name in module-level dicts
(pysteps/motion/interface.py:36-46 ``_methods``;
pysteps/extrapolation/interface.py:107-111 ``_extrapolation_methods``); there is
no entry-point discovery for them.  ``register()`` inserts the HIP callables under
new names, so that every caller that takes a method *name* - ``nowcasts.extrapolation``,
``nowcasts.steps`` (``extrap_method=``), sprog/anvil/linda/sseps, the blending
module, user scripts calling ``motion.get_method("LK_hip")`` - picks them up
unchanged.  With ``override=True`` the stock names ("semilagrangian", "lk",
"lucaskanade") are replaced as well (the reference's own identity tests
pysteps/tests/test_interfaces.py:69-78,220-233 then fail by design).
"""

FFT_NAME = "hip"
CASCADE_NAME = "fft_hip"  # pysteps.cascade.get_method("fft_hip") -> (decomposition_fft, recompose_fft)
NOISE_NAMES = {"parametric_hip": "parametric", "nonparametric_hip": "nonparametric"}
BPS_NAME = "bps_hip"  # vel_pert_method: the reference's generate_bps behind an initialiser that shares the unit fields
EXTRAPOLATION_NAMES = ("semilagrangian_hip",)
MOTION_NAMES = ("lk_hip", "lucaskanade_hip")
FEATURE_NAMES = {"blob_hip": "blob", "shitomasi_hip": "shitomasi"}  # pysteps.feature.get_method(...)
_STOCK_EXTRAPOLATION = ("semilagrangian",)
_STOCK_MOTION = ("lk", "lucaskanade")


def register_into(motion_methods, extrapolation_methods, override=False):
    """Insert the callables into the given dicts (either may be None). Returns the names added."""
    from .extrapolation.semilagrangian import extrapolate
    from .motion.lucaskanade import dense_lucaskanade

    added = []
    if extrapolation_methods is not None:
        for name in EXTRAPOLATION_NAMES + (_STOCK_EXTRAPOLATION if override else ()):
            extrapolation_methods[name] = extrapolate
            added.append("extrapolation:" + name)
    if motion_methods is not None:
        for name in MOTION_NAMES + (_STOCK_MOTION if override else ()):
            motion_methods[name] = dense_lucaskanade
            added.append("motion:" + name)
    return added


# pysteps modules that bind the generic nowcast loop by name (``from pysteps.nowcasts.utils import
# nowcast_main_loop``): steps.py:26, sprog.py, anvil.py, linda.py
_MAIN_LOOP_USERS = ("steps", "sprog", "anvil", "linda")


def _hip_aware_get_method(reference_get_method):
    """pysteps.utils.interface.get_method hard-codes the FFT method names (interface.py:240-243: an
    ``if name in ["numpy", "pyfftw", "scipy"]`` in front of the method dict), so a new FFT method
    cannot be added to a table: the lookup function itself is wrapped - ``"hip"`` is answered here,
    everything else goes to the reference function unchanged."""
    import functools  # noqa: PLC0415

    @functools.wraps(reference_get_method)
    def get_method(name="", **kwargs):
        if isinstance(name, str) and name.lower() == FFT_NAME:
            if "shape" not in kwargs:
                raise KeyError("mandatory keyword argument shape not given")  # interface.py:241-242
            from .utils.fft import get_hip  # noqa: PLC0415

            kwargs = dict(kwargs)
            return get_hip(kwargs.pop("shape"), **kwargs)
        return reference_get_method(name, **kwargs)

    get_method._pysteps_amd_reference = reference_get_method
    return get_method


def register_fft():
    """Make ``fft_method="hip"`` resolve (``pysteps.utils.get_method("hip", shape=...)``): the
    callers that take an FFT method name - nowcasts.steps / sseps / linda, the noise generators,
    the cascade decomposition - then run their transforms through csrc/fft.hip."""
    import pysteps.utils as utils_pkg  # noqa: PLC0415
    import pysteps.utils.interface as utils_if  # noqa: PLC0415

    if hasattr(utils_if.get_method, "_pysteps_amd_reference"):
        return []
    wrapped = _hip_aware_get_method(utils_if.get_method)
    utils_if.get_method = wrapped
    utils_pkg.get_method = wrapped
    try:  # the one module that binds the function by name (blending/utils.py:30)
        import pysteps.blending.utils as blending_utils  # noqa: PLC0415

        if hasattr(blending_utils, "utils_get_method"):
            blending_utils.utils_get_method = wrapped
    except Exception:
        pass
    return ["fft:" + FFT_NAME]


def register_spectral():
    """Insert the device cascade decomposition and noise generator into the reference's method tables
    (pysteps/cascade/interface.py:15-18 ``_cascade_methods``, pysteps/noise/interface.py:24-45
    ``_noise_methods``): ``decomp_method="fft_hip"`` and ``noise_method="nonparametric_hip"`` /
    ``"parametric_hip"`` (the reference's filter initialisation paired with the HIP generator)."""
    import pysteps.cascade.interface as cas_if  # noqa: PLC0415
    import pysteps.noise.interface as noise_if  # noqa: PLC0415

    from .cascade.decomposition import decomposition_fft, recompose_fft  # noqa: PLC0415
    from .noise.fftgenerators import generate_noise_2d_fft_filter  # noqa: PLC0415

    added = []
    cas_if._cascade_methods[CASCADE_NAME] = (decomposition_fft, recompose_fft)
    added.append("cascade:" + CASCADE_NAME)
    for name, stock in NOISE_NAMES.items():
        init = noise_if._noise_methods[stock][0]
        noise_if._noise_methods[name] = (init, generate_noise_2d_fft_filter)
        added.append("noise:" + name)
    from .noise.motion import initialize_bps  # noqa: PLC0415

    noise_if._noise_methods[BPS_NAME] = (initialize_bps, noise_if._noise_methods["bps"][1])
    added.append("noise:" + BPS_NAME)
    return added


def register_features(override=False):
    """Insert the HIP feature detectors into the reference's table (pysteps/feature/interface.py:26-29
    ``_detection_methods``) as ``"blob_hip"`` / ``"shitomasi_hip"`` (and under the stock names with ``override``):
    ``pysteps.feature.get_method("blob_hip")``, and through it ``dense_lucaskanade(fd_method="blob_hip")`` of the
    REFERENCE's Lucas-Kanade routine, then run the scale-space / corner kernels.  (``pysteps_amd``'s own
    ``dense_lucaskanade`` takes ``fd_method="blob"`` / ``"shitomasi"`` directly.)"""
    import pysteps.feature.interface as feat_if  # noqa: PLC0415

    from .feature import blob, shitomasi  # noqa: PLC0415

    added = []
    for name, stock in FEATURE_NAMES.items():
        fn = blob.detection if stock == "blob" else shitomasi.detection
        for key in (name,) + ((stock,) if override else ()):
            feat_if._detection_methods[key] = fn
            added.append("feature:" + key)
    return added


def unregister_fft():
    import pysteps.utils as utils_pkg  # noqa: PLC0415
    import pysteps.utils.interface as utils_if  # noqa: PLC0415

    ref = getattr(utils_if.get_method, "_pysteps_amd_reference", None)
    if ref is not None:
        utils_if.get_method = ref
        utils_pkg.get_method = ref


def patch_probmatching():
    """Replace ``pysteps.postprocessing.probmatching.nonparam_match_empirical_cdf`` by the device
    version.  The member loops reach it through the module attribute
    (``probmatching.nonparam_match_empirical_cdf(...)``: nowcasts/steps.py:1199, sprog.py:421,
    sseps.py:783,804, blending/steps.py:3333) and there is no method table for it, so the attribute is
    what has to change; the reference function stays reachable (calls with ``ignore_indices`` and
    inputs the device path declines are handed to it)."""
    import pysteps.postprocessing.probmatching as ref_mod  # noqa: PLC0415

    from .postprocessing import probmatching as hip_mod  # noqa: PLC0415

    if ref_mod.nonparam_match_empirical_cdf is hip_mod.nonparam_match_empirical_cdf:
        return []
    ref_mod._reference_nonparam_match_empirical_cdf = ref_mod.nonparam_match_empirical_cdf
    hip_mod._reference_fn = ref_mod.nonparam_match_empirical_cdf
    ref_mod.nonparam_match_empirical_cdf = hip_mod.nonparam_match_empirical_cdf
    return ["probmatching:nonparam_match_empirical_cdf"]


def unpatch_probmatching():
    """Undo :func:`patch_probmatching`."""
    import pysteps.postprocessing.probmatching as ref_mod  # noqa: PLC0415

    ref = getattr(ref_mod, "_reference_nonparam_match_empirical_cdf", None)
    if ref is not None:
        ref_mod.nonparam_match_empirical_cdf = ref
        del ref_mod._reference_nonparam_match_empirical_cdf


def patch_autoregression():
    """Replace ``pysteps.timeseries.autoregression.iterate_ar_model`` (reached through the module
    attribute: nowcasts/steps.py:1095,1137, sprog.py:398, sseps.py:678,749, anvil.py:483) by the device
    version; series it does not take run the reference's function."""
    import pysteps.timeseries.autoregression as ref_mod  # noqa: PLC0415

    from .timeseries import autoregression as hip_mod  # noqa: PLC0415

    if ref_mod.iterate_ar_model is hip_mod.iterate_ar_model:
        return []
    ref_mod._reference_iterate_ar_model = ref_mod.iterate_ar_model
    hip_mod._reference_fn = ref_mod.iterate_ar_model
    ref_mod.iterate_ar_model = hip_mod.iterate_ar_model
    return ["autoregression:iterate_ar_model"]


def unpatch_autoregression():
    """Undo :func:`patch_autoregression`."""
    import pysteps.timeseries.autoregression as ref_mod  # noqa: PLC0415

    ref = getattr(ref_mod, "_reference_iterate_ar_model", None)
    if ref is not None:
        ref_mod.iterate_ar_model = ref
        del ref_mod._reference_iterate_ar_model


def patch_dilated_mask():
    """Replace ``pysteps.nowcasts.utils.compute_dilated_mask`` (the incremental precipitation mask,
    reached as ``nowcast_utils.compute_dilated_mask(...)``: nowcasts/steps.py:983,1210, sseps.py:472,821)
    by the device version; masks it does not take run the reference's function."""
    import pysteps.nowcasts.utils as ref_mod  # noqa: PLC0415

    from .nowcasts import utils as hip_mod  # noqa: PLC0415

    if ref_mod.compute_dilated_mask is hip_mod.compute_dilated_mask:
        return []
    ref_mod._reference_compute_dilated_mask = ref_mod.compute_dilated_mask
    hip_mod._reference_dilated_mask = ref_mod.compute_dilated_mask
    ref_mod.compute_dilated_mask = hip_mod.compute_dilated_mask
    return ["nowcasts.utils:compute_dilated_mask"]


def unpatch_dilated_mask():
    """Undo :func:`patch_dilated_mask`."""
    import pysteps.nowcasts.utils as ref_mod  # noqa: PLC0415

    ref = getattr(ref_mod, "_reference_compute_dilated_mask", None)
    if ref is not None:
        ref_mod.compute_dilated_mask = ref
        del ref_mod._reference_compute_dilated_mask


def register(override=False, patch_main_loop=False, fft=True, probmatching=False, autoregression=False,
             dilated_mask=False):
    """Register with an importable pysteps; raises ImportError if pysteps is absent.

    ``patch_main_loop=True`` also installs the device-resident generic nowcast loop
    (:func:`pysteps_amd.nowcasts.utils.nowcast_main_loop`) in the nowcast modules: with
    ``extrap_method="semilagrangian_hip"`` all ensemble members are then advected by one kernel
    launch per time step and their trajectories stay in HBM; any other extrapolator runs exactly as
    before.  With that loop ``nowcasts.steps`` also runs its member update on device-resident state
    (:mod:`pysteps_amd.nowcasts.steps_resident`).  Parity level of that update: the default keeps the AR
    history as spectra and reproduces the reference's fields up to rounding, not bit for bit (identical NaN
    masks, no pixel decided differently by a threshold or a rank in the tests, 8e-8 relative L2 end to end);
    ``PYSTEPS_HIP_RESIDENT_DOMAIN=spatial`` selects the chain of spatial operators whose element-wise steps
    are bit-identical with the reference's.  :func:`unpatch_main_loop` restores the reference loop."""
    import pysteps.extrapolation.interface as ext_if  # noqa: PLC0415
    import pysteps.motion.interface as mot_if  # noqa: PLC0415

    added = register_into(mot_if._methods, ext_if._extrapolation_methods, override=override)
    try:
        added += register_features(override=override)
    except ImportError:
        pass  # pysteps.feature needs none of its optional dependencies at import time; a stripped-down install may lack it
    if fft:
        added += register_fft()
        added += register_spectral()
    if probmatching:
        added += patch_probmatching()
    if autoregression:
        added += patch_autoregression()
    if dilated_mask:
        added += patch_dilated_mask()
    if patch_main_loop:
        import importlib  # noqa: PLC0415

        from .nowcasts.utils import nowcast_main_loop  # noqa: PLC0415

        for name in _MAIN_LOOP_USERS:
            try:
                mod = importlib.import_module("pysteps.nowcasts." + name)
            except Exception:
                continue  # optional dependencies of that nowcast module are missing
            if hasattr(mod, "nowcast_main_loop"):
                if not hasattr(mod, "_reference_nowcast_main_loop"):
                    mod._reference_nowcast_main_loop = mod.nowcast_main_loop
                mod.nowcast_main_loop = nowcast_main_loop
                added.append("main_loop:" + name)
    return added


def unpatch_main_loop():
    """Undo ``register(patch_main_loop=True)``."""
    import importlib  # noqa: PLC0415

    for name in _MAIN_LOOP_USERS:
        try:
            mod = importlib.import_module("pysteps.nowcasts." + name)
        except Exception:
            continue
        ref = getattr(mod, "_reference_nowcast_main_loop", None)
        if ref is not None:
            mod.nowcast_main_loop = ref
            del mod._reference_nowcast_main_loop
