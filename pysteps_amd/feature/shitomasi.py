"""Shi-Tomasi corner detection on MI355X, drop-in for ``pysteps.feature.shitomasi.detection``
(reference: pysteps/feature/shitomasi.py:26-171).

The stand-alone form of the detector the dense Lucas-Kanade estimate uses (``pysteps_amd.motion.lucaskanade``):
mask buffer, min-max rescaling to uint8 and ``cv2.goodFeaturesToTrack`` (``useHarrisDetector=False``) as HIP kernels
(``csrc/lk.hip``: ``lk_stats1`` / ``lk_to_u8_bits`` / ``lk_corner_response_cols`` / ``lk_corner_select`` / ``corner_*``).
The Harris response is not implemented: such calls go to the reference when pysteps is importable.
"""

import warnings

import numpy as np

from ..device import DeviceArray

__all__ = ["detection"]


def detection(input_image, max_corners=1000, max_num_features=None, quality_level=0.01, min_distance=10, block_size=5,
              buffer_mask=5, use_harris=False, k=0.04, verbose=False, **kwargs):
    """Parameters and return value as documented for the reference (shitomasi.py:41-126): ``(p, 2)`` pixel
    coordinates (x, y) of the detected corners, strongest first."""
    from ..motion.lucaskanade import PreparedFrame, detect_corners  # noqa: PLC0415

    if input_image.ndim != 2:
        raise ValueError("input_image must be a two-dimensional array")
    supported = not use_harris and isinstance(block_size, (int, np.integer)) and 1 <= block_size <= 7 and block_size % 2 == 1
    if not supported:
        try:
            from pysteps.feature.shitomasi import detection as ref  # noqa: PLC0415
        except Exception as exc:
            raise NotImplementedError(
                "pysteps_amd shitomasi.detection: use_harris / block_size=%r is not implemented on the HIP path" % (block_size,)
            ) from exc
        if ref is detection or isinstance(input_image, DeviceArray):
            raise NotImplementedError("pysteps_amd shitomasi.detection: use_harris / block_size=%r" % (block_size,))
        warnings.warn("pysteps_amd shitomasi.detection: delegating to the reference CPU path")
        return ref(input_image, max_corners, max_num_features, quality_level, min_distance, block_size, buffer_mask,
                   use_harris, k, verbose, **kwargs)
    if isinstance(input_image, DeviceArray):
        frame = input_image
    else:
        arr = input_image
        dtype = np.float64 if np.asarray(arr).dtype == np.float64 else np.float32
        if isinstance(arr, np.ma.MaskedArray):
            arr = np.ma.filled(arr.astype(dtype, copy=True), np.nan)
        frame = DeviceArray.from_host(np.asarray(arr), dtype=dtype)
    prep = PreparedFrame(frame, 0, buffer_mask, want_features=True)
    points = detect_corners(prep, max_num_features if max_num_features is not None else max_corners, quality_level,
                            min_distance, block_size)
    if points.shape[0] == 0:
        points = np.empty(shape=(0, 2))  # shitomasi.py:166-167
    if verbose:
        print(f"--- {points.shape[0]} good features to track detected ---")
    return points
