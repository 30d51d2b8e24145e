"""Dense Lucas-Kanade optical flow on MI355X (HIP), drop-in for
``pysteps.motion.lucaskanade.dense_lucaskanade`` (reference:
pysteps/motion/lucaskanade.py:38-279).

Same signature, return values and exceptions.  Per frame the cleaning (NaN fill,
3x3 opening), the uint8 rescaling, the Shi-Tomasi response / selection, the
Gaussian pyramids, the Scharr gradients and the pyramidal tracker run as HIP
kernels (``csrc/lk.hip``, C ABI ``psh_lk_*``); the pooled sparse vectors (a few
thousand) are quality-controlled on the host exactly like the reference
(``pysteps_amd.utils.cleansing``) and interpolated to the grid by the IDW kernel
(``csrc/idw.hip``).  ``input_images`` may be a NumPy array / MaskedArray (result:
float64 ndarray like the reference) or a float32
:class:`pysteps_amd.device.DeviceArray` (result stays in HBM as float32).

The reference's default detector/interpolator pair (``fd_method="shitomasi"``, ``interp_method="idwinterp2d"``)
runs as ONE queued chain (``csrc/dense_lk.hip``); ``fd_method="blob"`` (``pysteps_amd.feature.blob``: scale-space
maxima, ``csrc/blob.hip``) and ``interp_method="rbfinterp2d"`` run stage by stage on the device; any other
``interp_method`` gets the HIP sparse stage followed by the reference's interpolation function; other choices
(``fd_method="tstorm"``, the Harris response) are delegated to the reference when pysteps is importable, else
NotImplementedError.
"""

import ctypes
import threading
import time
import warnings

import numpy as np

from .. import _lib
from ..device import DeviceArray
from ..utils.cleansing import decluster, detect_outliers_device
from ..utils.interpolate import idw_to_device, idwinterp2d

__all__ = ["dense_lucaskanade", "PreparedFrame", "detect_corners", "track_points"]

_N_STATS = 8
# set to False to run the stage-by-stage Python loop instead of psh_dense_lk_dev (same results;
# used by the tests to keep both orchestrations honest)
USE_NATIVE_ORCHESTRATION = True
# a resident motion field also as (m, n, 2) {u, v} pairs (DeviceArray.uv_pairs): the layout of the extrapolator's gather
# kernels (semilag_variant 7 / 5).  None (default): decided per field - the window kernel samples the planes, so the
# twin's stores are only spent for shapes it does not take (psh_semilag_window_shape: n < 96 or m < 64), where every
# long extrapolation call would otherwise interleave the planes again.
# True / False: always / never
WRITE_UV_TWIN = None
# one corner request (launch -> finish) is in flight per process: callers on several threads
# (the reference is re-entrant and gets called from dask workers) take turns here
_corner_lock = threading.Lock()


class PreparedFrame:
    """Device-side products of one input frame: cleaned float32 field, the two
    uint8 renderings (tracker / feature detector) and the statistics block."""

    def __init__(self, frame_dev, size_opening, buffer_mask, want_features):
        lib = _lib.lib()
        m, n = frame_dev.shape
        self.shape = (m, n)
        self.buffer_mask = int(buffer_mask)
        self.clean = DeviceArray((m, n), np.float32)
        self.track_u8 = DeviceArray((m, n), np.uint8)
        self.feature_u8 = DeviceArray((m, n), np.uint8) if want_features else None
        self.stats = DeviceArray((_N_STATS,), np.float32)
        # float64 frames are cleaned and quantised in double, like the reference does for them
        prepare = lib.psh_lk_prepare_f64_dev if frame_dev.dtype == np.float64 else lib.psh_lk_prepare_dev
        _lib.check(
            prepare(
                frame_dev.ptr, m, n, int(size_opening), self.buffer_mask, self.clean.ptr,
                self.track_u8.ptr, None if self.feature_u8 is None else self.feature_u8.ptr,
                self.stats.ptr,
            ),
            "psh_lk_prepare_dev",
        )


def detect_corners(prep, max_corners=1000, quality_level=0.01, min_distance=10, block_size=5):
    """Shi-Tomasi corners of a prepared frame -> (p,2) float32 (x,y) (shitomasi.py:153-171)."""
    lib = _lib.lib()
    m, n = prep.shape
    pts = np.empty((int(max_corners), 2), dtype=np.float32)
    count = ctypes.c_int(0)
    with _corner_lock:
        rc = lib.psh_lk_corners_dev(
            prep.feature_u8.ptr, prep.clean.ptr, prep.stats.ptr, m, n, int(block_size),
            prep.buffer_mask, float(quality_level), float(min_distance), int(max_corners),
            pts.ctypes.data, ctypes.byref(count),
        )
    _lib.check(rc, "psh_lk_corners_dev")
    return pts[: count.value].copy()


def _criteria(criteria):
    ctype, max_count, eps = criteria
    if not (ctype & 1):  # no COUNT bit: OpenCV substitutes 30 iterations
        max_count = 30
    if not (ctype & 2):  # no EPS bit: OpenCV substitutes 0.01
        eps = 0.01
    return int(max_count), float(eps)


def launch_corners(prep, max_corners=1000, quality_level=0.01, min_distance=10, block_size=5):
    """Queue the corner kernels of a prepared frame (asynchronous); pair with finish_corners."""
    m, n = prep.shape
    _lib.check(
        _lib.lib().psh_lk_corners_launch_dev(
            prep.feature_u8.ptr, prep.clean.ptr, prep.stats.ptr, m, n, int(block_size),
            prep.buffer_mask, float(quality_level), float(min_distance), int(max_corners),
        ),
        "psh_lk_corners_launch_dev",
    )
    return int(max_corners)


def finish_corners(max_corners):
    """Wait for the candidates, run the ordered min-distance pass -> (p,2) float32 (x,y)."""
    pts = np.empty((int(max_corners), 2), dtype=np.float32)
    count = ctypes.c_int(0)
    _lib.check(_lib.lib().psh_lk_corners_finish(pts.ctypes.data, ctypes.byref(count)), "psh_lk_corners_finish")
    return pts[: count.value].copy()


class PyramidPair:
    """Gaussian pyramids + Scharr gradients of a frame pair on the device (asynchronous build)."""

    def __init__(self, prev, nxt, winsize=(50, 50), nr_levels=3):
        m, n = prev.shape
        handle = ctypes.c_void_p()
        _lib.check(
            _lib.lib().psh_lk_pyramids_dev(prev.track_u8.ptr, nxt.track_u8.ptr, m, n, int(winsize[0]),
                                           int(winsize[1]), int(nr_levels), ctypes.byref(handle)),
            "psh_lk_pyramids_dev",
        )
        self._h = handle
        self._frames = (prev, nxt)  # level 0 lives in the prepared frames

    def track(self, points, criteria=(3, 10, 0), min_eig_thr=1e-4):
        p0 = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
        p1 = np.empty_like(p0)
        st = np.zeros(p0.shape[0], dtype=np.uint8)
        max_count, eps = _criteria(criteria)
        _lib.check(
            _lib.lib().psh_lk_track_pyr_dev(self._h, p0.ctypes.data, p0.shape[0], max_count, eps,
                                            float(min_eig_thr), p1.ctypes.data, st.ctypes.data),
            "psh_lk_track_pyr_dev",
        )
        return p1, st.astype(bool)

    def close(self):
        if self._h:
            _lib.load().psh_lk_pyramids_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def track_points(prev, nxt, points, winsize=(50, 50), nr_levels=3, criteria=(3, 10, 0),
                 min_eig_thr=1e-4):
    """Pyramidal LK between two prepared frames -> (next_points (p,2) f32, status (p,) bool)."""
    lib = _lib.lib()
    m, n = prev.shape
    p0 = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 2)
    p1 = np.empty_like(p0)
    st = np.zeros(p0.shape[0], dtype=np.uint8)
    max_count, eps = _criteria(criteria)
    _lib.check(
        lib.psh_lk_track_dev(
            prev.track_u8.ptr, nxt.track_u8.ptr, m, n, p0.ctypes.data, p0.shape[0],
            int(winsize[0]), int(winsize[1]), int(nr_levels), int(max_count), float(eps),
            float(min_eig_thr), p1.ctypes.data, st.ctypes.data,
        ),
        "psh_lk_track_dev",
    )
    return p1, st.astype(bool)


def _dense_lk_native(frames, on_device, dense, size_opening, buffer_mask, max_corners, quality_level,
                     min_distance, block_size, winsize, nr_levels, criteria, min_eig_thr,
                     nr_std_outlier, k_outlier, decl_scale, interp_kwargs):
    """psh_dense_lk_dev; None if the request is outside what that call takes (the stage-by-stage
    loop below then handles it)."""
    if not USE_NATIVE_ORCHESTRATION or set(interp_kwargs) - {"power", "k", "dist_offset", "nchunks", "hkey"}:
        return None
    if k_outlier is None or np.ndim(decl_scale) != 0:
        return None
    nr_fields, m, n = frames.shape
    max_count, eps = _criteria(criteria)
    k_idw = interp_kwargs.get("k", 20)
    prm = _lib.LkParams(
        int(size_opening), int(buffer_mask), int(max_corners), int(block_size),
        float(quality_level), float(min_distance), int(winsize[0]), int(winsize[1]), int(nr_levels),
        max_count, eps, float(min_eig_thr), float(nr_std_outlier), int(k_outlier), float(decl_scale),
        0 if k_idw is None else int(k_idw), float(interp_kwargs.get("power", 0.5)),
        float(interp_kwargs.get("dist_offset", 0.5)), 1 if frames.dtype == np.float64 else 0,
    )
    lib = _lib.lib()
    count = ctypes.c_int(0)
    if dense:
        field = DeviceArray((2, m, n), np.float32)
        # resident frames in, resident field out: the call only queues kernels (no count asked for,
        # so nothing waits for the device); host callers get the sample count with the field
        # (WRITE_UV_TWIN: a resident field also gets its {u, v}-interleaved twin, written by the interpolation kernel:
        # the layout the extrapolator's GATHER kernels sample, which would otherwise interleave the planes on every
        # call; the window kernel - the default - reads the planes, so the twin's 128 MB of stores are not spent)
        want_twin = (not lib.psh_semilag_window_shape(m, n)) if WRITE_UV_TWIN is None else bool(WRITE_UV_TWIN)
        pairs = DeviceArray((m, n, 2), np.float32) if on_device and want_twin else None
        rc = lib.psh_dense_lk_uv_dev(frames.ptr, nr_fields, m, n, ctypes.byref(prm), field.ptr,
                                     None if pairs is None else pairs.ptr, None, None, 0,
                                     None if on_device else ctypes.byref(count))
        if rc == _lib.PSH_EUNSUPPORTED:
            return None
        _lib.check(rc, "psh_dense_lk_uv_dev")
        if on_device:
            if pairs is not None:
                field.uv_pairs = pairs
            return field
        if count.value == 0:
            return np.zeros((2, m, n))
        return field.to_host().astype(np.float64)
    capacity = int(max_corners) * max(nr_fields - 1, 1)
    xy = np.empty((capacity, 2), dtype=np.float64)
    uv = np.empty((capacity, 2), dtype=np.float64)
    rc = lib.psh_dense_lk_dev(frames.ptr, nr_fields, m, n, ctypes.byref(prm), None, xy.ctypes.data,
                              uv.ctypes.data, capacity, ctypes.byref(count))
    if rc == _lib.PSH_EUNSUPPORTED:
        return None
    _lib.check(rc, "psh_dense_lk_dev")
    return xy[: count.value].copy(), uv[: count.value].copy()


_BLOB_KWARGS = {"max_num_features", "method", "threshold", "min_sigma", "max_sigma", "overlap", "num_sigma", "log_scale",
                "sigma_ratio"}


def _blob_points(prep, host_frame, fd_kwargs):
    """``feature.blob.detection`` on the cleaned frame -> (p, 2) float32 (x, y) (lucaskanade.py:230).  The reference
    detects on the frame in ITS dtype: for float64 host frames the cleaned frame is put together again in float64
    (the pixels the opening removed are those whose float32 rendering changed: they hold the frame's minimum)."""
    from ..feature.blob import detection  # noqa: PLC0415

    kwargs = {k: v for k, v in fd_kwargs.items() if k != "return_sigmas"}
    if host_frame is not None and np.asarray(host_frame).dtype == np.float64:
        orig = np.asarray(np.ma.filled(host_frame, np.nan) if isinstance(host_frame, np.ma.MaskedArray) else host_frame)
        clean32 = prep.clean.to_host()
        removed = np.isfinite(orig) & (clean32 != orig.astype(np.float32))
        image = np.where(removed, np.nanmin(orig), orig)
        image[~np.isfinite(orig)] = np.nan
    else:
        image = prep.clean
    return detection(image, **kwargs).astype(np.float32)


def _reference_dense_lk():
    try:
        from pysteps.motion.lucaskanade import dense_lucaskanade as ref  # noqa: PLC0415
    except Exception:
        return None
    return None if ref is dense_lucaskanade else ref


def _dense_with_reference_interpolator(input_images, lk_kwargs, fd_method, fd_kwargs, interp_method, interp_kwargs,
                                       nr_std_outlier, k_outlier, size_opening, decl_scale, verbose):
    """Another interpolation method: the sparse stage - features, tracking, outlier removal - runs on the HIP
    path, the vectors are declustered here and handed to the interpolation function, exactly as
    pysteps/motion/lucaskanade.py:199,264-274 does.  ``"rbfinterp2d"`` (pysteps/utils/interpolate.py:117-170, wraps
    scipy.interpolate.Rbf) is this package's: SciPy's weights, the grid evaluation on the device; any other name is
    looked up in the reference's table."""
    if isinstance(input_images, DeviceArray):
        raise NotImplementedError(
            "pysteps_amd dense_lucaskanade: interp_method=%r returns a host array; pass NumPy frames" % (interp_method,)
        )
    if interp_method == "rbfinterp2d":
        # SciPy's own weights, the grid evaluation on the device (utils/interpolate.py, csrc/rbf.hip)
        from ..utils.interpolate import rbfinterp2d as interpolation_method  # noqa: PLC0415
    else:
        try:
            from pysteps import utils as ref_utils  # noqa: PLC0415
        except Exception as exc:
            raise NotImplementedError(
                "pysteps_amd dense_lucaskanade: interp_method=%r needs pysteps' interpolation functions" % (interp_method,)
            ) from exc
        interpolation_method = ref_utils.get_method(interp_method)  # ValueError for unknown names, as the reference
    xy, uv = dense_lucaskanade(
        input_images, lk_kwargs, fd_method, fd_kwargs, "idwinterp2d", None, False, nr_std_outlier, k_outlier,
        size_opening, decl_scale, verbose,
    )
    domain_size = input_images.shape[1:]
    if xy.shape[0] == 0:  # lucaskanade.py:245-249
        return np.zeros((2, domain_size[0], domain_size[1]))
    if decl_scale > 1:  # :264-265
        xy, uv = decluster(xy, uv, decl_scale, 1, verbose)
    if xy.shape[0] == 0:  # :268-269
        return np.zeros((2, domain_size[0], domain_size[1]))
    xgrid = np.arange(domain_size[1])
    ygrid = np.arange(domain_size[0])
    return interpolation_method(xy, uv, xgrid, ygrid, **interp_kwargs)  # :272-274


def _frames_to_device(input_images):
    if isinstance(input_images, DeviceArray):
        if input_images.dtype not in (np.float32, np.float64):
            raise ValueError("device-resident input_images must be float32 or float64")
        return input_images, True
    arr = input_images
    # float64 frames (what pysteps passes as a rule) keep their precision: the reference cleans and
    # quantises in the dtype it is given; everything else is computed as float32
    dtype = np.float64 if np.asarray(arr).dtype == np.float64 else np.float32
    if isinstance(arr, np.ma.MaskedArray):
        arr = np.ma.filled(arr.astype(dtype, copy=True), np.nan)
    return DeviceArray.from_host(np.asarray(arr), dtype=dtype), False


def dense_lucaskanade(
    input_images,
    lk_kwargs=None,
    fd_method="shitomasi",
    fd_kwargs=None,
    interp_method="idwinterp2d",
    interp_kwargs=None,
    dense=True,
    nr_std_outlier=3,
    k_outlier=30,
    size_opening=3,
    decl_scale=20,
    verbose=False,
):
    """Run the Lucas-Kanade optical flow routine and interpolate the motion vectors.

    Parameters and returns as documented for the reference
    (pysteps/motion/lucaskanade.py:53-180): ``(2,m,n)`` motion field
    (x- and y-components, pixels per time step) or, with ``dense=False``, the
    sparse ``(xy, uv)`` arrays after outlier removal.
    """
    if input_images.ndim != 3:
        # check_input_frames (decorators.py:121-146)
        raise ValueError(
            "input_images dimension mismatch.\n"
            f"input_images.shape: {tuple(input_images.shape)}\n"
            "(t, x, y ) dimensions expected"
        )
    lk_kwargs = dict(lk_kwargs or {})
    fd_kwargs = dict(fd_kwargs or {})
    interp_kwargs = dict(interp_kwargs or {})

    unsupported = None
    fd_name = fd_method.lower() if isinstance(fd_method, str) else fd_method
    if fd_name not in ("shitomasi", "blob"):
        unsupported = "fd_method=%r" % (fd_method,)
    elif fd_name == "blob" and (fd_kwargs.get("method", "log") == "doh" or set(fd_kwargs) - _BLOB_KWARGS):
        unsupported = "fd_method='blob' with fd_kwargs %r" % (sorted(fd_kwargs),)
    elif fd_name == "shitomasi" and fd_kwargs.get("use_harris", False):
        unsupported = "use_harris=True"
    elif lk_kwargs.get("flags", 0) != 0:
        unsupported = "flags=%r" % (lk_kwargs.get("flags"),)
    elif size_opening not in (0, 3):
        unsupported = "size_opening=%r" % (size_opening,)
    else:
        # limits of the kernels (csrc/lk.hip kMaxBlockR / kMaxWin, csrc/idw.hip top-k registers):
        # checked HERE so that such calls reach the reference instead of failing in the library
        bs = fd_kwargs.get("block_size", 5) if fd_name == "shitomasi" else 5
        win = lk_kwargs.get("winsize", (50, 50))
        ik, ipow = interp_kwargs.get("k", 20), interp_kwargs.get("power", 0.5)
        if not (isinstance(bs, (int, np.integer)) and 1 <= bs <= 7 and bs % 2 == 1):
            unsupported = "block_size=%r (odd, <= 7 on the HIP path)" % (bs,)
        elif not (len(win) == 2 and 3 <= int(win[0]) <= 64 and 3 <= int(win[1]) <= 64):
            unsupported = "winsize=%r (3..64 on the HIP path)" % (win,)
        elif ik is not None and ik > 32:
            unsupported = "interp_kwargs k=%r (k <= 32 or k=None on the HIP path)" % (ik,)
        elif not ipow > 0:
            unsupported = "interp_kwargs power=%r (> 0 on the HIP path)" % (ipow,)

    if unsupported is not None:
        ref = _reference_dense_lk()
        if ref is None or isinstance(input_images, DeviceArray):
            raise NotImplementedError(
                "pysteps_amd dense_lucaskanade: %s is not implemented on the HIP path" % unsupported
            )
        warnings.warn("pysteps_amd dense_lucaskanade: %s -> delegating to the reference CPU path" % unsupported)
        return ref(input_images, lk_kwargs, fd_method, fd_kwargs, interp_method, interp_kwargs,
                   dense, nr_std_outlier, k_outlier, size_opening, decl_scale, verbose)

    if dense and interp_method != "idwinterp2d":
        return _dense_with_reference_interpolator(
            input_images, lk_kwargs, fd_name, fd_kwargs, interp_method, interp_kwargs, nr_std_outlier, k_outlier,
            size_opening, decl_scale, verbose,
        )

    if verbose:
        print("Computing the motion field with the Lucas-Kanade method.")
        t0 = time.time()

    frames, on_device = _frames_to_device(input_images)
    nr_fields, m, n = frames.shape

    max_corners = fd_kwargs.get("max_num_features") or fd_kwargs.get("max_corners", 1000)
    quality_level = fd_kwargs.get("quality_level", 0.01)
    min_distance = fd_kwargs.get("min_distance", 10)
    block_size = fd_kwargs.get("block_size", 5)
    buffer_mask = fd_kwargs.get("buffer_mask", 5)
    winsize = lk_kwargs.get("winsize", (50, 50))
    nr_levels = lk_kwargs.get("nr_levels", 3)
    criteria = lk_kwargs.get("criteria", (3, 10, 0))
    min_eig_thr = lk_kwargs.get("min_eig_thr", 1e-4)

    # ---- fast path: the whole estimate in one C-ABI call (csrc/dense_lk.hip) ----------------
    native = None if fd_name != "shitomasi" else _dense_lk_native(
        frames, on_device, dense, size_opening, buffer_mask, max_corners, quality_level, min_distance,
        block_size, winsize, nr_levels, criteria, min_eig_thr, nr_std_outlier, k_outlier, decl_scale,
        interp_kwargs,
    )
    if native is not None:
        if verbose:
            print("--- total time: %.2f seconds ---" % (time.time() - t0))
        return native

    prepared = [
        PreparedFrame(frames.view(t), size_opening, buffer_mask, want_features=fd_name == "shitomasi" and t < nr_fields - 1)
        for t in range(nr_fields)
    ]

    xy = np.empty(shape=(0, 2))
    uv = np.empty(shape=(0, 2))
    for t in range(nr_fields - 1):
        # corner kernels first, then the pyramids of the pair: the device builds them while
        # the host runs the ordered min-distance pass over the candidates
        if fd_name == "blob":
            # lucaskanade.py:230: the detector sees the frame after the opening, missing pixels still NaN
            pyramids = PyramidPair(prepared[t], prepared[t + 1], winsize, nr_levels)
            points = _blob_points(prepared[t], None if on_device else input_images[t], fd_kwargs)
        else:
            with _corner_lock:
                token = launch_corners(prepared[t], max_corners, quality_level, min_distance, block_size)
                pyramids = PyramidPair(prepared[t], prepared[t + 1], winsize, nr_levels)
                points = finish_corners(token)
            if fd_kwargs.get("verbose", False):
                print(f"--- {points.shape[0]} good features to track detected ---")
        if points.shape[0] == 0:
            pyramids.close()
            continue
        p1, st = pyramids.track(points, criteria, min_eig_thr)
        pyramids.close()
        if lk_kwargs.get("verbose", False):
            print(f"--- {int(st.sum())} sparse vectors found ---")
        if not st.any():
            continue
        xy = np.append(xy, points[st], axis=0)
        uv = np.append(uv, p1[st] - points[st], axis=0)

    def zero_field():
        if on_device:
            return DeviceArray((2, m, n), np.float32).fill_bytes(0)
        return np.zeros((2, m, n))

    if xy.shape[0] == 0:
        return zero_field() if dense else (xy, uv)

    outliers = detect_outliers_device(uv, nr_std_outlier, xy, k_outlier, verbose)
    xy, uv = xy[~outliers, :], uv[~outliers, :]
    if verbose:
        print("--- LK found %i sparse vectors ---" % xy.shape[0])
    if not dense:
        return xy, uv

    if decl_scale > 1:
        xy, uv = decluster(xy, uv, decl_scale, 1, verbose)
    if xy.shape[0] == 0:
        return zero_field()

    if on_device:
        nsamples = xy.shape[0]
        if nsamples == 1 or uv.max() == uv.min():  # trivial cases of the interpolator
            host = np.ones((2, m, n), dtype=np.float32) * uv[0].astype(np.float32)[:, None, None]
            uvgrid = DeviceArray.from_host(host)
        else:
            uvgrid = idw_to_device(
                xy, uv, m, n, power=interp_kwargs.get("power", 0.5), k=interp_kwargs.get("k", 20),
                dist_offset=interp_kwargs.get("dist_offset", 0.5),
            )
    else:
        uvgrid = idwinterp2d(xy, uv, np.arange(n), np.arange(m), **interp_kwargs)

    if verbose:
        print("--- total time: %.2f seconds ---" % (time.time() - t0))
    return uvgrid
