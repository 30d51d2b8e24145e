"""ctypes binding of libpysteps_hip.so (the C ABI declared in include/pysteps_hip.h).

There is deliberately no CPU fallback: if the shared object is missing or the
GPU cannot be initialised the product path raises, it never computes elsewhere.
"""

import ctypes
import os
import threading

from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
# PYSTEPS_HIP_LIB: another build of the library (the assertion build of `python -m pysteps_amd.build --debug`)
LIB_PATH = os.environ.get("PYSTEPS_HIP_LIB") or os.path.join(_PKG, "lib", "libpysteps_hip.so")

PSH_OK = 0
PSH_EINVAL = -1
PSH_EHIP = -2
PSH_ENOTINIT = -3
PSH_ENOMEM = -4
PSH_ECOMM = -5
PSH_EUNSUPPORTED = -6
PSH_EINPUT = -7


class LkParams(ctypes.Structure):
    """struct psh_lk_params of include/pysteps_hip.h."""

    _fields_ = [
        ("size_opening", c_int), ("buffer_mask", c_int), ("max_corners", c_int), ("block_size", c_int),
        ("quality_level", c_double), ("min_distance", c_double),
        ("win_w", c_int), ("win_h", c_int), ("max_level", c_int), ("max_count", c_int),
        ("epsilon", c_double), ("min_eig_threshold", c_double),
        ("nr_std_outlier", c_double), ("k_outlier", c_int),
        ("decl_scale", c_double),
        ("idw_k", c_int), ("idw_power", c_double), ("idw_dist_offset", c_double),
        ("frames_f64", c_int),
    ]


class HipLibraryError(RuntimeError):
    """libpysteps_hip.so is missing, failed to load, or a HIP/RCCL call failed."""


# name -> (restype, argtypes); kept in step with include/pysteps_hip.h
# (tests/test_capi_symbols.py parses the header and compares).
SIGNATURES = {
    "psh_init": (c_int, [c_int]),
    "psh_shutdown": (c_int, []),
    "psh_last_error": (c_char_p, []),
    "psh_version": (c_char_p, []),
    "psh_device_info": (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t), c_char_p, c_int]),
    "psh_set_option": (c_int, [c_char_p, c_int]),
    "psh_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "psh_free": (c_int, [c_void_p]),
    "psh_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "psh_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t]),
    "psh_memcpy_d2h_async": (c_int, [c_void_p, c_void_p, c_size_t]),
    "psh_convert_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_int]),
    "psh_axpy_f64_dev": (c_int, [c_void_p, c_void_p, ctypes.c_double, c_size_t]),
    "psh_count_above_dev": (c_int, [c_void_p, c_size_t, c_double, POINTER(c_double), POINTER(c_double)]),
    "psh_host_alloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "psh_host_free": (c_int, [c_void_p]),
    "psh_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t]),
    "psh_memset": (c_int, [c_void_p, c_int, c_size_t]),
    "psh_sync": (c_int, []),
    "psh_event_create": (c_int, [POINTER(c_void_p)]),
    "psh_event_destroy": (c_int, [c_void_p]),
    "psh_event_record": (c_int, [c_void_p]),
    "psh_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "psh_rbf_eval_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_int,
                                 c_double, c_void_p]),
    "psh_fft_rfft2_dev": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "psh_fft_irfft2_dev": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "psh_fft_irfft2_min_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "psh_fft_c2c2_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "psh_cascade_decompose_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                          POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "psh_cascade_decompose_stats_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "psh_cascade_recompose_dev": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(c_double), POINTER(c_double),
                                          c_double, c_void_p]),
    "psh_noise_filter_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "psh_probmatch_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "psh_probmatch_async_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "psh_probmatch_status": (c_int, [c_int]),
    "psh_order_statistic_dev": (c_int, [c_void_p, c_size_t, c_size_t, POINTER(c_double)]),
    "psh_probmatch_plan_create": (c_int, [c_void_p, c_size_t, c_void_p]),
    "psh_probmatch_plan_destroy": (c_int, [c_void_p]),
    "psh_probmatch_planned_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "psh_blob_cube_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "psh_blob_peaks_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "psh_blob_gather_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "psh_steps_mask_probmatch_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "psh_dilated_mask_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "psh_steps_incremental_mask_dev": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_int, c_int, c_int, c_void_p]),
    "psh_ar_iterate_dev": (c_int, [c_void_p, c_int, c_size_t, POINTER(c_double), c_int, c_void_p, c_void_p]),
    "psh_steps_ar_recompose_dev": (c_int, [c_void_p, c_int, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]),
    "psh_steps_ar_recompose_raw_dev": (c_int, [c_void_p, c_int, c_int, c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_void_p]),
    "psh_steps_spectral_sums_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "psh_steps_spectral_ar_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "psh_steps_phase_ar_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "psh_mask_row_offsets_dev": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "psh_expand_compact_c128_dev": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "psh_field_min_key_dev": (c_int, [c_void_p, c_size_t, c_void_p]),
    "psh_steps_mask_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "psh_steps_mean_shift_dev": (c_int, [c_void_p, c_size_t, c_double, c_double]),
    "psh_ge_mask_dev": (c_int, [c_void_p, c_size_t, c_double, c_void_p]),
    "psh_nan_where_dev": (c_int, [c_void_p, c_void_p, c_size_t]),
    "psh_lerp_dev": (c_int, [c_void_p, c_void_p, c_double, c_void_p, c_size_t]),
    "psh_rng_create": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_void_p)]),
    "psh_rng_randn_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_int]),
    "psh_rng_uniform_dev": (c_int, [c_void_p, c_size_t, c_double, c_double, c_void_p, c_int]),
    "psh_rng_wait": (c_int, [c_void_p]),
    "psh_rng_check": (c_int, [c_void_p]),
    "psh_rng_get_state": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "psh_rng_destroy": (c_int, [c_void_p]),
    "psh_lk_greedy_host": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_int, c_void_p, c_void_p]),
    "psh_lk_order_host": (c_int, [c_void_p, c_int, c_float, c_double, c_int, c_int, c_double, c_int, c_void_p, c_void_p]),
    "psh_idw_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_int, c_double, c_double, c_double, c_void_p]),
    "psh_idw_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_int, c_double, c_double, c_void_p]),
    "psh_lk_prepare_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "psh_lk_prepare_f64_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "psh_lk_corners_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_int, c_void_p, POINTER(c_int)]),
    "psh_lk_track_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_double, c_void_p, c_void_p]),
    "psh_comm_unique_id_bytes": (c_int, []),
    "psh_comm_unique_id": (c_int, [c_void_p]),
    "psh_comm_init": (c_int, [c_void_p, c_int, c_int]),
    "psh_comm_broadcast": (c_int, [c_void_p, c_size_t, c_int]),
    "psh_comm_allgather": (c_int, [c_void_p, c_void_p, c_size_t]),
    "psh_comm_allreduce_f32": (c_int, [c_void_p, c_size_t, c_int]),
    "psh_comm_destroy": (c_int, []),
    "psh_outliers_local_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_void_p]),
    "psh_decluster_host": (c_int, [c_void_p, c_void_p, c_int, c_double, c_int, c_void_p, c_void_p, POINTER(c_int)]),
    "psh_vectors_finish_host": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_double, c_int, c_int, c_void_p, c_void_p,
                                        POINTER(c_int), POINTER(c_int), c_void_p, c_void_p]),
    "psh_velocity_unit_dev": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "psh_members_pack_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "psh_semilag_members_packed_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "psh_semilag_members_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "psh_semilag_members_state_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "psh_members_state_to_disp_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "psh_members_disp_to_state_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "psh_lk_corners_launch_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_double, c_double, c_int]),
    "psh_lk_corners_finish": (c_int, [c_void_p, POINTER(c_int)]),
    "psh_lk_pyramids_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "psh_lk_pyramids_free": (c_int, [c_void_p]),
    "psh_lk_pyramids_band": (c_int, [c_void_p, c_int, c_int, POINTER(c_int)]),  # handle, frame_rows, band_first_row, top
    "psh_lk_band_stats_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "psh_lk_band_open_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "psh_lk_band_to_u8_dev": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "psh_lk_band_response_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "psh_lk_band_select_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_double, c_void_p, c_void_p, c_int, c_void_p]),
    "psh_lk_track_pyr_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p]),
    "psh_db_transform_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_double, c_double, c_int]),
    "psh_nonfinite_count_f64_dev": (c_int, [c_void_p, c_size_t, POINTER(c_double)]),
    "psh_field_stats_dev": (c_int, [c_void_p, c_size_t, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "psh_dense_lk_dev": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(LkParams), c_void_p, c_void_p, c_void_p, c_int, POINTER(c_int)]),
    "psh_dense_lk_uv_dev": (c_int, [c_void_p, c_int, c_int, c_int, POINTER(LkParams), c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    POINTER(c_int)]),
    "psh_semilag_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "psh_semilag_uv_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int,
                                   c_void_p]),
    "psh_semilag_window_shape": (c_int, [c_int, c_int]),
    "psh_semilag_kernel": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "psh_semilag_rows_dev": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "psh_semilag_host": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_int, POINTER(c_int)]),
}

_lock = threading.RLock()
_lib = None
_initialised = False


def load():
    """dlopen the library and declare the prototypes (no GPU needed)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                "libpysteps_hip.so not found at %s - build it with "
                "`python -m pysteps_amd.build` (hipcc, gfx950). pysteps_amd has no CPU fallback."
                % LIB_PATH
            )
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as exc:
            raise HipLibraryError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise HipLibraryError("%s does not export %s" % (LIB_PATH, name)) from exc
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
        return lib


def last_error():
    msg = load().psh_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what=""):
    """Map a C status code onto the exception the reference would raise."""
    if rc == PSH_OK:
        return
    msg = "%s%s" % (what + ": " if what else "", last_error())
    if rc in (PSH_EINVAL, PSH_EINPUT):
        raise ValueError(msg)
    if rc == PSH_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == PSH_ENOMEM:
        raise MemoryError(msg)
    raise HipLibraryError(msg)


def default_device():
    """Device index for this process: PYSTEPS_HIP_DEVICE, else LOCAL_RANK, else 0."""
    for key in ("PYSTEPS_HIP_DEVICE", "LOCAL_RANK"):
        val = os.environ.get(key)
        if val not in (None, ""):
            return int(val)
    return 0


def lib():
    """Loaded library with the GPU bound (psh_init done); raises if there is no GPU."""
    global _initialised
    handle = load()
    if not _initialised:
        with _lock:
            if not _initialised:
                check(handle.psh_init(default_device()), "psh_init")
                _initialised = True
    return handle
