"""Run the REAL reference ``dense_lucaskanade`` (oracle/_ref) around a stand-in ``cv2`` module whose five functions
are the restated OpenCV algorithms of oracle/lk_opencv.py, and compare with the oracle's own pipeline.

What this pins: everything of rows a4 / a5 that is pysteps' own code - NaN masking and minimum fill, the `> minimum`
field of the opening, the uint8 renderings, the buffer mask and its row quirk (shitomasi.py:140), argument order and
defaults of the five cv2 call sites, pooling, outlier test, declustering, interpolation - against the reference's code
itself.  What it cannot pin: the five OpenCV algorithms (the stand-in IS the restatement).
Prints one JSON line.  A process of its own: the reference decides at import time whether cv2 exists."""
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import lk_opencv as lk  # noqa: E402

calls = {"getStructuringElement": 0, "morphologyEx": 0, "dilate": 0, "goodFeaturesToTrack": 0, "calcOpticalFlowPyrLK": 0}
cv2 = types.ModuleType("cv2")
cv2.MORPH_ELLIPSE, cv2.MORPH_OPEN = 2, 2
cv2.TERM_CRITERIA_EPS, cv2.TERM_CRITERIA_COUNT = 2, 1


def get_structuring_element(shape, ksize):
    calls["getStructuringElement"] += 1
    assert shape == cv2.MORPH_ELLIPSE and ksize[0] == ksize[1]
    return lk._structuring_element_ellipse(int(ksize[0])).astype(np.uint8)


def morphology_ex(src, op, kernel):
    calls["morphologyEx"] += 1
    assert op == cv2.MORPH_OPEN and src.dtype == np.uint8
    k = kernel.astype(bool)
    return lk._morph(lk._morph(src.astype(bool), k, True), k, False).astype(np.uint8)


def dilate(src, kernel, iterations=1):
    calls["dilate"] += 1
    assert src.dtype == np.uint8 and kernel.shape[0] == kernel.shape[1] and kernel.all() and iterations == 1
    return lk.dilate_mask(src.astype(bool), int(kernel.shape[0])).astype(np.uint8)


def good_features_to_track(image, mask=None, maxCorners=0, qualityLevel=0.01, minDistance=1, blockSize=3,
                           useHarrisDetector=False, k=0.04):
    calls["goodFeaturesToTrack"] += 1
    assert image.dtype == np.uint8 and not useHarrisDetector
    pts = lk.good_features_to_track(image, mask.astype(bool), max_corners=int(maxCorners), quality=qualityLevel,
                                    min_distance=minDistance, block_size=int(blockSize))
    return None if pts.shape[0] == 0 else pts[:, None, :]


def calc_optical_flow_pyr_lk(prev, nxt, p0, p1, winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01), flags=0,
                             minEigThreshold=1e-4):
    calls["calcOpticalFlowPyrLK"] += 1
    assert prev.dtype == np.uint8 and nxt.dtype == np.uint8 and p1 is None and flags == 0
    got, status = lk.calc_optical_flow_pyr_lk(prev, nxt, np.asarray(p0, dtype=np.float32).reshape(-1, 2), win=tuple(winSize),
                                              max_level=int(maxLevel), max_count=int(criteria[1]), epsilon=float(criteria[2]),
                                              min_eig_threshold=minEigThreshold)
    return got.reshape(np.shape(p0)), status.astype(np.uint8)[:, None], None


cv2.getStructuringElement, cv2.morphologyEx, cv2.dilate = get_structuring_element, morphology_ex, dilate
cv2.goodFeaturesToTrack, cv2.calcOpticalFlowPyrLK = good_features_to_track, calc_optical_flow_pyr_lk
sys.modules["cv2"] = cv2
build_ref.activate()
from pysteps.motion.lucaskanade import dense_lucaskanade as ref_dense  # noqa: E402
from tools import synth  # noqa: E402

report = {}
for name, (m, n, nframes, nan) in {"plain": (96, 128, 2, False), "nan_three_frames": (120, 100, 3, True)}.items():
    frames = synth.steps_frames(m, n, nframes).astype(np.float64)
    if nan:
        frames[:, :7, :] = np.nan
        frames[:, 40:52, 60:75] = np.nan
    kw = dict(fd_kwargs=dict(max_corners=300, min_distance=6, block_size=5, buffer_mask=5),
              lk_kwargs=dict(winsize=(21, 21), nr_levels=2))
    want_xy, want_uv = ref_dense(frames.copy(), dense=False, verbose=False, **kw)
    want = ref_dense(frames.copy(), verbose=False, **kw)
    okw = dict(max_corners=300, min_distance=6, block_size=5, buffer_mask=5, winsize=(21, 21), nr_levels=2)
    got_xy, got_uv = lk.dense_lucaskanade(frames.copy(), dense=False, **okw)
    got = lk.dense_lucaskanade(frames.copy(), **okw)
    report[name] = {
        "vectors": int(want_xy.shape[0]),
        "sparse_equal": bool(want_xy.shape == got_xy.shape and np.array_equal(want_xy, got_xy) and np.array_equal(want_uv, got_uv)),
        "dense_max_abs_diff": float(np.max(np.abs(np.asarray(want) - got))),
        "dense_scale": float(np.max(np.abs(want))),
    }
report["calls"] = calls
if len(sys.argv) > 1:
    # (for the GPU test: frames and what the reference made of them with its default options, to hold the device
    # path against the reference's own orchestration directly)
    frames = synth.steps_frames(192, 224, 3).astype(np.float64)
    frames[:, 100:120, :30] = np.nan
    xy, uv = ref_dense(frames.copy(), dense=False, verbose=False)
    field = ref_dense(frames.copy(), verbose=False)
    np.savez(sys.argv[1], frames=frames, xy=xy, uv=uv, field=np.asarray(field))
print(json.dumps(report))
