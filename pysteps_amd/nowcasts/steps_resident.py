"""The STEPS member update with every member's state resident in HBM (the loop).

``pysteps.nowcasts.steps`` advances its ensemble once per time step through
``StepsNowcaster.__update_state`` (pysteps/nowcasts/steps.py:1057-1219): per member white noise from
the member's ``RandomState`` -> noise filter -> cascade decomposition -> AR(p) step per cascade level ->
recomposition -> precipitation mask -> probability matching -> mask update -> domain mask.  The
reference keeps all of that in NumPy arrays inside the ``state`` dictionary; with the device operators
patched in one by one (round 2) every call still crossed PCIe twice.

:class:`ResidentSteps` is that update as ONE chain of kernel launches on device-resident state:

* the members' cascades - per member and level a ring of ``ar_order`` planes (``x_new`` overwrites the
  oldest plane instead of ``np.concatenate``-ing the series up by one);
* the members' random streams (:class:`pysteps_amd.noise.randstate.DeviceRandomStates`: MT19937 + polar
  method, the very numbers ``randstate.randn`` would return), drawn one time step ahead on their own
  stream, beside the rest of the update;
* band-pass weights, noise filter, CDF-matching target, grey-scale masks.

It is built from the ``state`` / ``params`` dictionaries the reference hands to ``nowcast_main_loop``
(steps.py:1014-1055) - the drop-in point is the loop (``register(patch_main_loop=True)``), not a fork
of the nowcaster - and :func:`try_create` declines (returns None, the reference's own update function
runs) whenever an option is set that the chain does not implement: ``use_full_fft`` filters, no noise, grid
sides beyond the FFT kernels (any side up to 4096 is fine, powers of two up to 8192), more than 16 cascade
levels or AR order above 8.  ``mask_method`` None /
``"incremental"`` / ``"obs"`` / ``"sprog"`` and ``probmatching_method`` None / ``"cdf"`` / ``"mean"`` are
implemented, and so is the reference's own ``domain="spectral"`` (round 4: the state is spectral THERE - per level
the coefficients where the band-pass weight exceeds 1e-12 -, the noise a field of unit phasors with phases from
``RandomState.uniform``; `csrc/steps_loop.hip` ``spectral_phase_ar``, one irfft2 per member update).  A failure
while the device state is being built (out of memory, a generator the device streams do not reproduce)
also ends in the reference's update, with a warning.

Parity - what a caller who patches the main loop gets.  The DEFAULT update keeps the AR history as spectra
(two transforms per member update instead of nine; the level variances come from Parseval's identity):
its fields are those of the reference's spatial chain up to rounding - not bit for bit: held in
tests/test_steps_resident_gpu.py to identical NaN masks, at most 1e-5 of the pixels decided differently by
a threshold or a rank (observed: none), median difference <= 1e-14 of the field's range, generator states
equal, and end to end to 1e-4 relative L2 against the stock ``nowcasts.steps`` (observed 8e-8).
``PYSTEPS_HIP_RESIDENT_DOMAIN=spatial`` selects the chain of the reference's spatial operators instead,
whose element-wise steps are the reference's arithmetic operation by operation (csrc/steps_loop.hip,
bit-identical with ``iterate_ar_model`` + ``recompose_fft``).  In both the transforms agree with numpy.fft
to ~1e-16 and the random stream is NumPy's up to the rounding of ``log`` (csrc/cr_log.h).  With the reference's
``domain="spectral"`` the phases are NumPy's bit for bit, ``cos`` / ``sin`` come from the device library and the
two standard deviations of the noise are taken as the constants they are (|exp(i theta)| = 1): the same bars.
"""

import ctypes
import os

import numpy as np

from .. import _lib
from ..cascade.decomposition import _device_weights
from ..device import DeviceArray
from ..noise.randstate import DeviceRandomStates
from ..utils import fft as hip_fft

__all__ = ["ResidentSteps", "try_create", "recognises"]

_MAX_LEVELS, _MAX_ORDER = 16, 8


def _is_fn(obj, module_suffix, name):
    """``obj`` is the function ``name`` of a module ending in ``module_suffix`` (the reference's or ours)."""
    return callable(obj) and getattr(obj, "__name__", "") == name and getattr(obj, "__module__", "").endswith(module_suffix)


def _c_doubles(values):
    arr = np.ascontiguousarray(values, dtype=np.float64)
    return arr, arr.ctypes.data_as(ctypes.c_void_p)


def recognises(func):
    """True if ``func`` is the update function of the reference's STEPS nowcaster
    (``StepsNowcaster.__update_state``, bound; pysteps/nowcasts/steps.py:439-457)."""
    owner = getattr(func, "__self__", None)
    return owner is not None and type(owner).__name__ == "StepsNowcaster" and getattr(func, "__name__", "") == "__update_state"


def _percentile_index(count, pct):
    """Index into the sorted field that compute_percentile_mask (nowcasts/utils.py:129-135) thresholds at:
    ``argmin |x - pct|`` over ``x[k] = 1.0 * (count - k) / count`` (first minimum), evaluated on a window around
    the analytic position with NumPy's own operations; ties between equal VALUES do not change the threshold.
    None where the reference reads one element past the end (index count - 1)."""
    if not 0.0 <= pct <= 1.0 or count < 2:
        return None
    k0 = int(round(count - pct * count))
    ks = np.arange(max(0, k0 - 4), min(count, k0 + 5))
    x = 1.0 * (count - ks) / count
    i = int(ks[np.argmin(np.abs(x - pct))])
    # |x - pct| falls towards the minimum and rises after it: the window holds the global first minimum
    return None if i >= count - 1 else i


def _spectral_std_of_moduli(x, m, n):
    """``pysteps.utils.spectral.std(X, (m, n))`` (utils/spectral.py:231-238) for a half spectrum whose MODULI are the
    real array ``x``: the phase noise of the spectral domain has unit modulus, so the standard deviations of the
    filtered noise and of its cascade levels are functions of the filters alone."""
    res = np.sum(x ** 2) - x[0, 0] ** 2
    res += np.sum(x[:, 1:] ** 2) if n % 2 == 1 else np.sum(x[:, 1:-1] ** 2)
    return np.sqrt(res / (m * n) ** 2)


def _self_conjugate_columns_symmetric(planes, n):
    """The spectral form of the update reads level variances off the spectrum (Parseval), which needs every product
    ``spectrum x weights`` to stay Hermitian: a real filter has to take the same value at (ky, kx) and (-ky, kx) on the
    two columns of an rfft2 half spectrum that are their own mirror images (kx = 0 and, for even n, the Nyquist column).
    Filters that are functions of |k| - every band-pass filter and noise filter pysteps builds - are; anything else
    keeps the chain of spatial operators.  ``planes``: (..., m, n // 2 + 1) NumPy array; DeviceArrays are trusted."""
    if isinstance(planes, DeviceArray):
        return True
    a = np.asarray(planes)
    if a.ndim < 2 or a.shape[-1] != n // 2 + 1 or np.iscomplexobj(a):
        return False
    cols = [0] + ([a.shape[-1] - 1] if n % 2 == 0 else [])
    for c in cols:
        col = a[..., :, c]
        mirrored = np.concatenate([col[..., :1], col[..., :0:-1]], axis=-1)
        scale = float(np.max(np.abs(col))) if col.size else 0.0
        if not np.allclose(col, mirrored, rtol=1e-9, atol=1e-12 * scale):
            return False
    return True


def try_create(func, state, params, shape, n_updates):
    """A :class:`ResidentSteps` for the update function ``func`` of the reference's STEPS nowcaster, or
    None if ``func`` is something else or uses options outside the resident chain."""
    if not recognises(func):
        return None
    try:
        return ResidentSteps(state, params, shape, n_updates)
    except _Declined:
        return None
    except (RuntimeError, MemoryError, ValueError) as exc:
        # the device could not take the state (out of memory, a random generator the device streams do not
        # reproduce, a HIP error while uploading): whatever was allocated is released with the half-built
        # object and the reference's own update runs - a nowcast never fails because the fast path declined late.
        # PYSTEPS_HIP_STRICT=1 (the test suite sets it): a failure while the state is built is an error, so that a
        # programming mistake in the constructor cannot hide behind the fallback as a mere slowdown
        import gc
        import os
        import warnings

        if os.environ.get("PYSTEPS_HIP_STRICT", "0") not in ("", "0"):
            raise

        gc.collect()
        warnings.warn("pysteps_amd: the resident STEPS update is not used (%s: %s); the reference's update runs"
                      % (type(exc).__name__, exc), RuntimeWarning, stacklevel=2)
        return None


class _Declined(Exception):
    pass


class ResidentSteps:
    def __init__(self, state, params, shape, n_updates):
        need = ("noise_method", "domain", "generate_noise", "pert_gen", "decomp_method", "recomp_method", "filter", "phi",
                "noise_std_coeffs", "n_cascade_levels", "n_ens_members", "mask_method", "probmatching_method", "precip",
                "precip_thr", "domain_mask")
        if not isinstance(params, dict) or not isinstance(state, dict) or any(k not in params for k in need):
            raise _Declined
        m, n = (int(s) for s in shape)
        p = params
        if p["noise_method"] is None or p["domain"] not in ("spatial", "spectral") or not hip_fft.supported_shape((m, n)):
            raise _Declined
        self.ref_spectral = p["domain"] == "spectral"  # the reference's own spectral domain (steps.py:122-126)
        if not _is_fn(p["generate_noise"], "noise.fftgenerators", "generate_noise_2d_fft_filter"):
            raise _Declined
        if not _is_fn(p["decomp_method"], "cascade.decomposition", "decomposition_fft"):
            raise _Declined
        if not _is_fn(p["recomp_method"], "cascade.decomposition", "recompose_fft"):
            raise _Declined
        F = p["pert_gen"]
        if not isinstance(F, dict) or F.get("use_full_fft", True) or tuple(F.get("input_shape", ())) != (m, n):
            raise _Declined
        if np.shape(F["field"]) != (m, n // 2 + 1) or np.any(~np.isfinite(F["field"])):
            raise _Declined  # the reference raises its ValueError on its own path
        if p["mask_method"] not in (None, "incremental", "obs", "sprog") or p["probmatching_method"] not in (None, "cdf", "mean"):
            raise _Declined
        self.B, self.L = int(p["n_ens_members"]), int(p["n_cascade_levels"])
        phi = np.asarray(p["phi"], dtype=np.float64)
        if phi.ndim != 2 or phi.shape[0] != self.L:
            raise _Declined
        self.p = phi.shape[1] - 1
        weights = p["filter"]["weights_2d"]
        if (self.L > _MAX_LEVELS or not 1 <= self.p <= _MAX_ORDER or np.shape(weights) != (self.L, m, n // 2 + 1)
                or len(p["filter"]["weights_1d"]) != self.L):
            raise _Declined
        gens = state.get("randgen_prec")
        cascades = state.get("precip_cascades")
        decomp = state.get("precip_decomp")
        if gens is None or len(gens) != self.B or cascades is None or decomp is None or len(decomp) != self.B:
            raise _Declined
        if not isinstance(cascades, DeviceArray) and len(cascades) != self.B:
            raise _Declined
        resident_cascades = isinstance(cascades, DeviceArray)  # state that never was on the host (bench.py)
        if resident_cascades and (self.ref_spectral or cascades.shape != (self.B, self.L, self.p, m, n)
                                  or cascades.dtype != np.float64):
            raise _Declined
        level_masks = None
        if self.ref_spectral:
            # decomposition.py:233-236: level k lives on the coefficients where its weight exceeds 1e-12
            level_masks = np.asarray(weights) > 1e-12
            counts = [int(level_masks[k].sum()) for k in range(self.L)]
        for j in range(self.B):
            if self.ref_spectral:
                d = decomp[j]
                if (not d.get("normalized", False) or d.get("domain") != "spectral" or not d.get("compact_output", False)
                        or np.shape(d.get("weight_masks")) != (self.L, m, n // 2 + 1)
                        or not np.array_equal(np.asarray(d["weight_masks"], dtype=bool), level_masks)):
                    raise _Declined
                if len(cascades[j]) != self.L or any(np.shape(cascades[j][k]) != (self.p, counts[k]) for k in range(self.L)):
                    raise _Declined
                continue
            if not resident_cascades and (len(cascades[j]) != self.L or any(np.shape(c) != (self.p, m, n) for c in cascades[j])):
                raise _Declined
            if not decomp[j].get("normalized", False) or decomp[j].get("domain") != "spatial":
                raise _Declined

        # ---- every option that can still decline is looked at BEFORE anything is allocated or uploaded ----
        if p["mask_method"] == "incremental":
            masks = state.get("mask_prec")
            struct = np.asarray(p.get("struct")) if p.get("struct") is not None else None
            rim = p.get("mask_rim")
            if (masks is None or struct is None or struct.ndim != 2 or rim is None or not 0 <= int(rim) <= 254
                    or int((struct != 0).sum()) > 1024 or p["precip_thr"] is None):
                raise _Declined
            if isinstance(masks, DeviceArray):
                if masks.shape != (self.B, m, n) or masks.dtype != np.float64:
                    raise _Declined
            elif len(masks) != self.B or any(np.shape(mk) != (m, n) for mk in masks):
                raise _Declined
        elif p["mask_method"] == "obs":
            if state.get("mask_prec") is None or np.shape(state["mask_prec"]) != (m, n):
                raise _Declined
        elif p["mask_method"] == "sprog":
            det, det_d = state.get("precip_m"), state.get("precip_m_d")
            if (det is None or not isinstance(det_d, dict) or len(det) != self.L or not det_d.get("normalized", False)
                    or p.get("war") is None or _percentile_index(m * n, float(p["war"])) is None):
                raise _Declined
            if self.ref_spectral:
                if (det_d.get("domain") != "spectral" or not det_d.get("compact_output", False)
                        or any(np.shape(det[k]) != (phi.shape[1] - 1, counts[k]) for k in range(self.L))
                        or not np.array_equal(np.asarray(det_d.get("weight_masks"), dtype=bool), level_masks)):
                    raise _Declined
            elif det_d.get("domain") != "spatial" or any(np.shape(c) != (phi.shape[1] - 1, m, n) for c in det):
                raise _Declined
        if p["probmatching_method"] == "cdf":
            tgt = p["precip"]
            if tgt is None or (isinstance(tgt, DeviceArray) and (tgt.shape != (m, n) or tgt.dtype != np.float64)) \
                    or (not isinstance(tgt, DeviceArray) and np.shape(tgt) != (m, n)):
                raise _Declined
        elif p["probmatching_method"] == "mean" and p.get("mu_0") is None:
            raise _Declined

        self._lib = _lib.lib()
        self.m, self.n, self.plane = m, n, m * n
        self.params, self.state = p, state
        self.n_updates, self.done = int(n_updates), 0
        self.phi, self._phi_p = _c_doubles(phi)
        self.noise_std, self._noise_std_p = _c_doubles(np.asarray(p["noise_std_coeffs"], dtype=np.float64).reshape(self.L))
        self.mu = [np.ascontiguousarray(decomp[j]["means"], dtype=np.float64) for j in range(self.B)]
        self.sigma = [np.ascontiguousarray(decomp[j]["stds"], dtype=np.float64) for j in range(self.B)]
        self.weights = _device_weights(weights)
        self.noise_filter = _device_weights(F["field"])
        # The AR history as SPECTRA (default): the update is linear between the white noise and the recomposed
        # field apart from two standardisations that need second moments only (Parseval) - two transforms per
        # member update instead of nine (csrc/steps_loop.hip).  PYSTEPS_HIP_RESIDENT_DOMAIN=spatial keeps the
        # level fields and the chain of the reference's spatial operators (bit-identical element-wise part).
        self.spectral = (not self.ref_spectral and os.environ.get("PYSTEPS_HIP_RESIDENT_DOMAIN", "spectral") != "spatial"
                         and _self_conjugate_columns_symmetric(weights, n) and _self_conjugate_columns_symmetric(F["field"], n))
        nc = n // 2 + 1
        # AR history: (B, L, p, m, n); slot s of the ring holds x[s] of the reference's series at start.  With the
        # spectral form a level field only passes through ONE staging plane on its way to its spectrum: the
        # device never holds the spatial history and its spectra together (8 B L p m n bytes less at the peak).
        if self.ref_spectral:
            # the reference's compact arrays, scattered into full half-spectrum planes on the device (the level's mask
            # is read off the resident weights); the two standard deviations of the noise (fftgenerators.py:435-437, decomposition.py:
            # 219-220 through utils/spectral.py:208-238) only see |exp(i theta)| = 1: constants of the nowcast
            spectra = DeviceArray((self.B, self.L, self.p, m, nc), np.complex128)
            stage = DeviceArray((self.p, max(counts)), np.complex128)
            offsets = DeviceArray((self.L, m + 1), np.int32)
            wplane = m * nc * 8
            for k in range(self.L):
                _lib.check(self._lib.psh_mask_row_offsets_dev(self.weights.ptr + k * wplane, m, nc, offsets.view(k).ptr),
                           "psh_mask_row_offsets_dev")

            def expand(levels, dst_ptr):  # levels: L compact arrays (p, counts[k]) -> (L, p, m, nc) planes at dst_ptr
                for k in range(self.L):
                    src = np.ascontiguousarray(levels[k], dtype=np.complex128)
                    _lib.check(self._lib.psh_memcpy_h2d(stage.ptr, src.ctypes.data, src.nbytes), "h2d")
                    for slot in range(self.p):
                        _lib.check(self._lib.psh_expand_compact_c128_dev(self.weights.ptr + k * wplane, m, nc, offsets.view(k).ptr,
                                                                         stage.ptr + slot * counts[k] * 16,
                                                                         dst_ptr + (k * self.p + slot) * m * nc * 16),
                                   "psh_expand_compact_c128_dev")
                    _lib.check(self._lib.psh_sync(), "sync")  # `src` may go, the staging block is reused

            for j in range(self.B):
                expand(cascades[j], spectra.ptr + j * self.L * self.p * m * nc * 16)
            self._expand_compact = expand
            self.cascades = spectra
            self.field_spec = DeviceArray((m, nc), np.complex128)

            def spectral_std(x):
                return _spectral_std_of_moduli(x, m, n)

            f0 = np.array(F["field"], dtype=np.float64)
            f0[0, 0] = 0.0
            self.inv_std_noise = 1.0 / spectral_std(f0)  # (NumPy divides a complex array by a real scalar this way)
            f0 *= self.inv_std_noise
            w = np.asarray(weights, dtype=np.float64)
            self.inv_std_levels, self._inv_std_levels_p = _c_doubles([1.0 / spectral_std(f0 * w[k]) for k in range(self.L)])
        elif self.spectral:
            spectra = DeviceArray((self.B, self.L, self.p, m, nc), np.complex128)
            stage = None if resident_cascades else DeviceArray((m, n), np.float64)
            for j in range(self.B):
                for k in range(self.L):
                    for slot in range(self.p):
                        q = (j * self.L + k) * self.p + slot
                        if resident_cascades:
                            src_ptr = cascades.ptr + q * self.plane * 8
                        else:
                            src = np.ascontiguousarray(cascades[j][k][slot], dtype=np.float64)
                            _lib.check(self._lib.psh_memcpy_h2d(stage.ptr, src.ctypes.data, src.nbytes), "h2d")
                            src_ptr = stage.ptr
                        _lib.check(self._lib.psh_fft_rfft2_dev(src_ptr, m, n, spectra.ptr + q * m * nc * 16), "psh_fft_rfft2_dev")
                        if not resident_cascades:
                            _lib.check(self._lib.psh_sync(), "sync")  # `src` and the staging plane are reused
            self.cascades = spectra  # (the level fields are not kept)
            self.noise_spec = DeviceArray((m, nc), np.complex128)
            self.field_spec = DeviceArray((m, nc), np.complex128)
            self.level_sums = DeviceArray((16,), np.float64)
        elif resident_cascades:
            self.cascades = cascades
        else:
            self.cascades = DeviceArray((self.B, self.L, self.p, m, n), np.float64)
            stride = self.p * self.plane * 8
            for j in range(self.B):
                for k in range(self.L):
                    src = np.ascontiguousarray(cascades[j][k], dtype=np.float64)
                    _lib.check(self._lib.psh_memcpy_h2d(self.cascades.ptr + (j * self.L + k) * stride, src.ctypes.data, src.nbytes), "h2d")
            _lib.check(self._lib.psh_sync(), "sync")
        self.head = 0  # slot of the oldest entry
        self.thr = float(p["precip_thr"]) if p["precip_thr"] is not None else None
        self.mask_method, self.pm_method = p["mask_method"], p["probmatching_method"]
        self.grey = self.keep = None
        if self.mask_method == "incremental":
            masks = state["mask_prec"]
            self.struct = np.ascontiguousarray(np.asarray(p["struct"]) != 0, dtype=np.uint8)
            self.rim = int(p["mask_rim"])
            self._bit_mask = os.environ.get("PYSTEPS_HIP_BIT_MASK", "1") != "0"  # (development switch)
            if self.struct.ndim != 2 or not 0 <= self.rim <= 254 or int(self.struct.sum()) > 1024 or self.thr is None:
                raise _Declined
            if isinstance(masks, DeviceArray):
                if masks.shape != (self.B, m, n) or masks.dtype != np.float64:
                    raise _Declined
                self.grey = masks
            else:
                if len(masks) != self.B or any(np.shape(mk) != (m, n) for mk in masks):
                    raise _Declined
                self.grey = DeviceArray((self.B, m, n), np.float64)
                for j, mk in enumerate(masks):  # member by member: no stacked host copy of 8 m n B bytes
                    src = np.ascontiguousarray(mk, dtype=np.float64)
                    _lib.check(self._lib.psh_memcpy_h2d(self.grey.ptr + j * self.plane * 8, src.ctypes.data, src.nbytes), "h2d")
                _lib.check(self._lib.psh_sync(), "sync")
            self.wet = DeviceArray((m, n), np.uint8)
        elif self.mask_method == "obs":
            self.keep = DeviceArray.from_host(np.ascontiguousarray(state["mask_prec"], dtype=np.uint8))
        elif self.mask_method == "sprog":
            # steps.py:1089-1114: the deterministic AR(p) model of the S-PROG field runs beside the members, its
            # recomposed field thresholded at the percentile that keeps the observed wet-area ratio is the
            # mask of EVERY member at that time step
            det, det_d = state.get("precip_m"), state.get("precip_m_d")
            self.war = float(p["war"])
            self.det_index = _percentile_index(self.plane, self.war)
            if self.det_index is None:
                raise _Declined  # (the reference's own index arithmetic runs off the end there)
            if self.ref_spectral:
                self.det = DeviceArray((self.L, self.p, m, nc), np.complex128)
                self._expand_compact(det, self.det.ptr)
            else:
                self.det = DeviceArray.from_host(np.ascontiguousarray(np.stack([np.asarray(c, dtype=np.float64) for c in det])))
            self.det_head = 0
            self.det_mu = np.ascontiguousarray(det_d["means"], dtype=np.float64)
            self.det_sigma = np.ascontiguousarray(det_d["stds"], dtype=np.float64)
            self.det_field = DeviceArray((m, n), np.float64)
            self.keep = DeviceArray((m, n), np.uint8)
        self._expand_compact = None  # (releases the staging block of the compact uploads)
        self.target = None
        if self.pm_method == "cdf":
            tgt = p["precip"]
            if isinstance(tgt, DeviceArray):
                if tgt.shape != (m, n) or tgt.dtype != np.float64:
                    raise _Declined
                self.target = tgt
            else:
                self.target = DeviceArray.from_host(np.ascontiguousarray(tgt, dtype=np.float64))
        elif self.pm_method == "mean":
            self.mu_0 = float(p["mu_0"])
        dm = p["domain_mask"]
        self.domain_mask = None
        if dm is not None and np.any(dm):
            self.domain_mask = DeviceArray.from_host(np.ascontiguousarray(dm, dtype=np.uint8))
        # CDF matching without a wait per member: the fields before the matching are kept for one time step
        # and every member's outcome lands in a device word (checked once, at the end of the update)
        # ... against a plan of the target: the observation is the same for every member and time step
        self.pm_plan = None
        if self.pm_method == "cdf":
            import ctypes  # noqa: PLC0415

            handle = ctypes.c_void_p()
            _lib.check(self._lib.psh_probmatch_plan_create(self.target.ptr, self.plane, ctypes.byref(handle)),
                       "psh_probmatch_plan_create")
            self.pm_plan = handle
        self.pre = DeviceArray((self.B, m, n), np.float64) if self.pm_method == "cdf" else None
        self.pm_status = DeviceArray((max(self.B, 2),), np.int32) if self.pm_method == "cdf" else None
        self.min_key = DeviceArray((8,), np.uint64)
        self.eps = self.eps_stats = self.noise = None
        if not (self.spectral or self.ref_spectral):  # the chain of spatial operators only
            self.eps = DeviceArray((self.L, m, n), np.float64)  # cascade of one member's noise field
            self.eps_stats = DeviceArray((self.L, 2), np.float64)  # (mean, std) of its levels
            self.noise = DeviceArray((m, n), np.float64)
        if self.ref_spectral:  # phases (m, n/2+1) per member instead of white noise (m, n)
            self.rng = DeviceRandomStates(gens, m * nc, n_draws=min(self.n_updates, 4096))
            self.white = [DeviceArray((self.B, m, nc), np.float64), DeviceArray((self.B, m, nc), np.float64)]
        else:
            self.rng = DeviceRandomStates(gens, self.plane, n_draws=min(self.n_updates, 4096))
            self.white = [DeviceArray((self.B, m, n), np.float64), DeviceArray((self.B, m, n), np.float64)]
        self._white_ready = False

    # ------------------------------------------------------------------
    def _draw(self, slot):
        if self.ref_spectral:  # fftgenerators.py:407: theta = randstate.uniform(low=0.0, high=2.0 * np.pi, size=(m, n/2+1))
            self.rng.uniform(0.0, 2.0 * np.pi, self.m, self.n // 2 + 1, out=self.white[slot], side=True)
        else:
            self.rng.randn(self.m, self.n, out=self.white[slot], side=True)

    def update(self):
        """One ``__update_state``: float64 DeviceArray ``(n_members, m, n)`` of the members' new fields."""
        lib, m, n, plane = self._lib, self.m, self.n, self.plane
        slot = self.done & 1
        # the draws of the time steps whose fields the caller has read back are complete: a short one
        # (stale values in the tail of its buffer) stops the nowcast here, not at finish()
        self.rng.check()
        if not self._white_ready:
            self._draw(slot)
        self.rng.wait()
        white = self.white[slot]
        # the NEXT time step's draw starts now, on the generators' own stream, beside this whole update: its
        # buffer was last read by the previous update (queued long ago); producing the words of six 4096^2
        # fields takes 30 ms - queued behind the last member's noise filter (the first version) it ran
        # beside a sixth of the update and the next update waited for the rest
        if self.done + 1 < self.n_updates:
            self._draw(slot ^ 1)
            self._white_ready = True
        else:
            self._white_ready = False
        out = DeviceArray((self.B, m, n), np.float64)
        lvl_stride = self.p * plane * 8
        if self.mask_method == "sprog":
            self._update_sprog_mask()
        for j in range(self.B):
            result = out.view(j)
            field = self.pre.view(j) if self.pre is not None else result  # what the matching reads
            if self.ref_spectral:
                # phases -> unit phasors -> filter, standardisation, levels, AR step, recomposition per coefficient; ONE transform
                nc = n // 2 + 1
                casc = self.cascades.ptr + j * self.L * self.p * m * nc * 16
                _lib.check(lib.psh_steps_phase_ar_dev(casc, self.L, self.p, m, n, self.head, self._phi_p, white.view(j).ptr,
                                                      self.noise_filter.ptr, self.weights.ptr, self.inv_std_noise,
                                                      self._inv_std_levels_p, self._noise_std_p, self.mu[j].ctypes.data,
                                                      self.sigma[j].ctypes.data, self.field_spec.ptr), "psh_steps_phase_ar_dev")
                # (the transform's last pass also takes the field's minimum: no sweep of its own)
                _lib.check(lib.psh_fft_irfft2_min_dev(self.field_spec.ptr, m, n, field.ptr, self.min_key.ptr), "psh_fft_irfft2_min_dev")
            elif self.spectral:
                # rfft2(white); level variances by Parseval; AR step + recomposition on the spectra; ONE inverse transform
                nc = n // 2 + 1
                casc = self.cascades.ptr + j * self.L * self.p * m * nc * 16
                _lib.check(lib.psh_fft_rfft2_dev(white.view(j).ptr, m, n, self.noise_spec.ptr), "psh_fft_rfft2_dev")
                _lib.check(lib.psh_steps_spectral_sums_dev(self.noise_spec.ptr, self.noise_filter.ptr, self.weights.ptr, self.L, m, n,
                                                           self.level_sums.ptr), "psh_steps_spectral_sums_dev")
                _lib.check(lib.psh_steps_spectral_ar_dev(casc, self.L, self.p, m, n, self.head, self._phi_p, self.noise_spec.ptr,
                                                         self.noise_filter.ptr, self.weights.ptr, self.level_sums.ptr,
                                                         self._noise_std_p, self.mu[j].ctypes.data, self.sigma[j].ctypes.data,
                                                         self.field_spec.ptr), "psh_steps_spectral_ar_dev")
                # (the transform's last pass also takes the field's minimum: no sweep of its own)
                _lib.check(lib.psh_fft_irfft2_min_dev(self.field_spec.ptr, m, n, field.ptr, self.min_key.ptr), "psh_fft_irfft2_min_dev")
            else:
                # fftgenerators.py:400-433 (the filter part), decomposition.py:77-262 with normalize=True
                _lib.check(lib.psh_noise_filter_dev(white.view(j).ptr, self.noise_filter.ptr, m, n, self.noise.ptr), "psh_noise_filter_dev")
                # (the levels stay unnormalised: the AR kernel standardises them on the way in, one sweep less)
                _lib.check(lib.psh_cascade_decompose_stats_dev(self.noise.ptr, self.weights.ptr, self.L, m, n, self.eps.ptr, self.eps_stats.ptr),
                           "psh_cascade_decompose_stats_dev")
                # steps.py:1116-1146 + 1176-1185
                _lib.check(
                    lib.psh_steps_ar_recompose_raw_dev(self.cascades.ptr + j * self.L * lvl_stride, self.L, self.p, plane, self.head,
                                                       self._phi_p, self.eps.ptr, self.eps_stats.ptr, self._noise_std_p,
                                                       self.mu[j].ctypes.data, self.sigma[j].ctypes.data, field.ptr, self.min_key.ptr),
                    "psh_steps_ar_recompose_raw_dev")
            masked = self.grey is not None or self.keep is not None
            if masked and self.pm_method == "cdf" and field.ptr != result.ptr:
                # steps.py:1221-1240 + 1198-1201: the matching's first sweep (statistics of the initial array) applies
                # the mask on its way - one pass over the field less than the two calls below; the outcome is read
                # after the last member
                _lib.check(lib.psh_steps_mask_probmatch_dev(
                    self.pm_plan, field.ptr, plane, None if self.grey is None else self.grey.view(j).ptr,
                    None if self.grey is not None else self.keep.ptr, self.min_key.ptr, result.ptr, self.pm_status.ptr + 4 * j),
                    "psh_steps_mask_probmatch_dev")
                self._finish_member(j, result)
                continue
            if self.grey is not None:  # steps.py:1221-1240
                _lib.check(lib.psh_steps_mask_dev(field.ptr, plane, self.grey.view(j).ptr, None, self.min_key.ptr), "psh_steps_mask_dev")
            elif self.keep is not None:
                _lib.check(lib.psh_steps_mask_dev(field.ptr, plane, None, self.keep.ptr, self.min_key.ptr), "psh_steps_mask_dev")
            if self.pm_method == "cdf":  # steps.py:1198-1201; the outcome is read after the last member
                _lib.check(lib.psh_probmatch_planned_dev(self.pm_plan, field.ptr, plane, result.ptr, self.pm_status.ptr + 4 * j),
                           "psh_probmatch_planned_dev")
            elif self.pm_method == "mean":  # steps.py:1203-1206
                _lib.check(lib.psh_steps_mean_shift_dev(result.ptr, plane, self.thr, self.mu_0), "psh_steps_mean_shift_dev")
            self._finish_member(j, result)
        if self.pm_method == "cdf":
            # ONE wait per time step: a member the bucket pass declined (thousands of tied wet values) is
            # matched by the reference's function from its kept field, its mask update is redone
            status = self.pm_status.to_host()
            self.rng.check()  # (this time step's draw is behind that wait)
            for j in range(self.B):
                if status[j] == 0:
                    continue
                rc = lib.psh_probmatch_status(int(status[j]))
                if rc != _lib.PSH_EUNSUPPORTED:
                    _lib.check(rc, "psh_probmatch_dev")
                matched = self._probmatch_on_host(self.pre.view(j))
                _lib.check(lib.psh_memcpy_d2d(out.view(j).ptr, matched.ptr, plane * 8), "d2d")
                self._finish_member(j, out.view(j))
        self.head = (self.head + 1) % self.p
        self.done += 1
        return out

    def _finish_member(self, j, field):
        """steps.py:1209-1217: the member's incremental mask from its new field, the domain mask."""
        lib, m, n, plane = self._lib, self.m, self.n, self.plane
        if self.grey is not None:
            kr = self.struct.ctypes.data_as(ctypes.c_void_p)
            kh, kw = int(self.struct.shape[0]), int(self.struct.shape[1])
            # threshold + dilations on bit masks in two kernels; structures that entry point does not take (no centre
            # element, a rim wider than its tiles' halo) go through the byte masks
            rc = lib.psh_steps_incremental_mask_dev(field.ptr, m, n, self.thr, kr, kh, kw, self.rim, self.grey.view(j).ptr) \
                if self._bit_mask else _lib.PSH_EUNSUPPORTED
            if rc == _lib.PSH_EUNSUPPORTED:
                self._bit_mask = False
                _lib.check(lib.psh_ge_mask_dev(field.ptr, plane, self.thr, self.wet.ptr), "psh_ge_mask_dev")
                _lib.check(lib.psh_dilated_mask_dev(self.wet.ptr, m, n, kr, kh, kw, self.rim, self.grey.view(j).ptr),
                           "psh_dilated_mask_dev")
            else:
                _lib.check(rc, "psh_steps_incremental_mask_dev")
        if self.domain_mask is not None:
            _lib.check(lib.psh_nan_where_dev(field.ptr, self.domain_mask.ptr, plane), "psh_nan_where_dev")

    def _update_sprog_mask(self):
        """steps.py:1089-1114 `__update_deterministic_ar_model` + nowcasts/utils.py:102-138 `compute_percentile_mask`:
        AR step of the deterministic cascade (no noise), recomposition, and the mask `field >= s[i]` with s the
        sorted field and i the index whose exceedance fraction is closest to the wet-area ratio."""
        import ctypes  # noqa: PLC0415

        lib, plane = self._lib, self.plane
        if self.ref_spectral:  # the same AR step without innovation on the compact spectral levels, one transform
            _lib.check(lib.psh_steps_phase_ar_dev(self.det.ptr, self.L, self.p, self.m, self.n, self.det_head, self._phi_p, None, None,
                                                  self.weights.ptr, 0.0, None, None, self.det_mu.ctypes.data,
                                                  self.det_sigma.ctypes.data, self.field_spec.ptr), "psh_steps_phase_ar_dev")
            _lib.check(lib.psh_fft_irfft2_dev(self.field_spec.ptr, self.m, self.n, self.det_field.ptr), "psh_fft_irfft2_dev")
        else:
            _lib.check(lib.psh_steps_ar_recompose_dev(self.det.ptr, self.L, self.p, plane, self.det_head, self._phi_p, None, None,
                                                      self.det_mu.ctypes.data, self.det_sigma.ctypes.data, self.det_field.ptr, None),
                       "psh_steps_ar_recompose_dev")
        self.det_head = (self.det_head + 1) % self.p
        thr = ctypes.c_double()
        rc = lib.psh_order_statistic_dev(self.det_field.ptr, plane, self.det_index, ctypes.byref(thr))
        if rc == _lib.PSH_EUNSUPPORTED:  # a value plateau the bucket pass declines: the reference's sort
            from pysteps.nowcasts.utils import compute_percentile_mask  # noqa: PLC0415

            mask = compute_percentile_mask(self.det_field.to_host(), self.war)
            self.keep = DeviceArray.from_host(np.ascontiguousarray(mask, dtype=np.uint8))
            return
        _lib.check(rc, "psh_order_statistic_dev")
        _lib.check(lib.psh_ge_mask_dev(self.det_field.ptr, plane, thr.value, self.keep.ptr), "psh_ge_mask_dev")

    def _probmatch_on_host(self, field):
        """The device CDF matching declined (thousands of tied wet values, infinities in the target):
        this one call goes through the reference's function and comes back."""
        from ..postprocessing.probmatching import _reference  # noqa: PLC0415

        got = _reference()(field.to_host(), np.asarray(self.target.to_host(), dtype=np.float64))
        return DeviceArray.from_host(np.ascontiguousarray(got, dtype=np.float64))

    def finish(self):
        """Hand the generators back to the host ``RandomState`` objects (whatever the caller draws next
        continues the same streams) and release the device state."""
        self.rng.sync_back()
        self.rng.close()
        self._release_plan()
        self.cascades = self.eps = self.white = self.grey = None

    def abort(self):
        """The loop ended by an exception: join the generators' own stream BEFORE the noise buffers go back to
        the block cache (a queued draw may still be writing them: a later allocation of that size would receive
        a buffer that is being filled), hand the host generators their streams back if the device still
        answers, release the device state.  Never raises.

        The host generators come back ONE draw ahead of the last completed update when the loop was not at its
        end: update() queues the next time step's draw beside the update it runs (``_draw(slot ^ 1)``), and a
        MT19937 stream cannot be wound back.  A caller that catches the exception and goes on drawing from the
        same ``RandomState`` objects continues a valid stream, but not the one the reference would continue -
        re-seed (or keep copies made with ``get_state()`` before the nowcast) where stream parity matters."""
        try:
            if getattr(self, "rng", None) is not None:
                try:
                    self.rng.wait()
                    _lib.check(self._lib.psh_sync(), "psh_sync")
                    self.rng.sync_back()
                except Exception:
                    pass
                self.rng.close()  # (waits for the generators' stream)
        except Exception:
            pass
        try:
            self._release_plan()
        except Exception:
            pass
        self.cascades = self.eps = self.white = self.grey = None

    def _release_plan(self):
        if getattr(self, "pm_plan", None):
            self._lib.psh_probmatch_plan_destroy(self.pm_plan)
            self.pm_plan = None

    def __del__(self):
        try:
            self._release_plan()
        except Exception:
            pass

    def cascade_levels(self, j):
        """Member j's latest cascade level fields as the reference holds them (L, m, n) - test hook."""
        newest = (self.head + self.p - 1) % self.p
        if self.spectral:
            nc = self.n // 2 + 1
            out = DeviceArray((self.L, self.m, self.n), np.float64)
            for k in range(self.L):
                at = ((j * self.L + k) * self.p + newest) * self.m * nc * 16
                _lib.check(self._lib.psh_fft_irfft2_dev(self.cascades.ptr + at, self.m, self.n, out.view(k).ptr), "psh_fft_irfft2_dev")
            return out.to_host()
        got = self.cascades.to_host()
        return got[j, :, newest]
