"""Capture OpenCV golden vectors for the Lucas-Kanade front end (LK parity is UNPINNED
because OpenCV is absent from every box this repository has been built on).

Run this ONCE on any machine that has ``cv2`` and a pysteps checkout (``/root/reference``, an installed
pysteps, or ``oracle/_ref``):

    python tools/make_lk_golden_cv2.py [--pysteps /path/to/pysteps/checkout]

It runs the REAL reference on the seeded synthetic frames of ``tools/synth.py`` - the five cv2 call
sites (``utils/images.py:72,75`` getStructuringElement + morphologyEx, ``feature/shitomasi.py:137``
dilate, ``:162`` goodFeaturesToTrack, ``tracking/lucaskanade.py:164-171`` calcOpticalFlowPyrLK) and the
whole ``dense_lucaskanade`` - and writes ``tests/golden/lk_opencv.npz``:

* per case the input frames, the opened frame (``morph_opening``), the two uint8 renderings the
  reference hands to OpenCV (recorded by wrapping the cv2 functions: they are not returned by the
  reference), the corner list in goodFeaturesToTrack's order, the tracked vectors and status,
  the sparse vectors and the dense field of ``dense_lucaskanade``;
* ``cv2_version`` / ``pysteps_version`` of the run.

``tests/test_lk_cv2_golden.py`` switches on when that file exists: the oracle (oracle/lk_opencv.py) and
the device path are then compared with OpenCV itself - uint8 renderings, opening and corner list bit
for bit, vectors within 1e-2 px, dense field within 1e-3 - and DESIGN.md's "PARITY UNPINNED" can go.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "lk_opencv.npz")

# (name, shape, frames, NaN pattern, seed)
CASES = [
    ("rain_256", (256, 256), 2, None, 11),
    ("rain_nan_300x260", (300, 260), 3, "corner+speck", 12),
    ("rain_512x384", (512, 384), 2, "border", 13),
    ("rain_1024", (1024, 1024), 2, None, 14),
]


def make_frames(shape, count, nan, seed):
    """dB rain frames: the base field shifted by (2, -1) px per frame + smooth evolution."""
    from scipy.ndimage import gaussian_filter

    from tools import synth

    m, n = shape
    base = synth.rain_field_db(m, n, seed=seed, sigma=max(m / 64.0, 2.0))
    frames = []
    for t in range(count):
        f = np.roll(base, (2 * t, -t), axis=(0, 1)).copy()
        e = gaussian_filter(np.random.default_rng(100 * seed + t).standard_normal((m, n)), 2.0)
        f = np.where(f > -15.0, f + 0.5 * e / e.std(), -15.0).astype(np.float32)
        frames.append(f)
    frames = np.stack(frames)
    if nan == "corner+speck":
        frames[:, : m // 5, : n // 3] = np.nan
        frames[:, m // 2, n // 2] = np.nan
    elif nan == "border":
        frames[:, synth.border_nan_mask(m, n, 0.1)] = np.nan
    return frames


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pysteps", default=None, help="directory that contains the pysteps package (default: importable "
                    "pysteps, else /root/reference, else oracle/_ref)")
    args = ap.parse_args()
    import cv2  # noqa: F401 - fails loudly here if OpenCV is missing: that is the point of this script

    for cand in (args.pysteps, None, "/root/reference", os.path.join(ROOT, "oracle", "_ref")):
        if cand:
            sys.path.insert(0, cand)
        try:
            import pysteps  # noqa: F401

            break
        except Exception:
            if cand:
                sys.path.remove(cand)
    else:
        raise SystemExit("pysteps is not importable; pass --pysteps")
    from pysteps.feature import shitomasi
    from pysteps.motion.lucaskanade import dense_lucaskanade
    from pysteps.tracking import lucaskanade as tracking
    from pysteps.utils import images

    recorded = {}
    real_gftt, real_lk = cv2.goodFeaturesToTrack, cv2.calcOpticalFlowPyrLK

    def gftt(image, *a, **k):
        recorded["feature_u8"] = np.array(image, copy=True)
        recorded["feature_mask"] = np.array(k.get("mask"), copy=True) if k.get("mask") is not None else None
        return real_gftt(image, *a, **k)

    def pyrlk(prev, nxt, p0, p1, **k):
        recorded["prev_u8"], recorded["next_u8"] = np.array(prev, copy=True), np.array(nxt, copy=True)
        out = real_lk(prev, nxt, p0, p1, **k)
        recorded["lk_p1"], recorded["lk_status"] = np.array(out[0], copy=True), np.array(out[1], copy=True)
        return out

    store = {"cv2_version": np.array(cv2.__version__), "pysteps_version": np.array(getattr(pysteps, "__version__", "?")),
             "cases": np.array([c[0] for c in CASES])}
    cv2.goodFeaturesToTrack, cv2.calcOpticalFlowPyrLK = gftt, pyrlk
    try:
        for name, shape, count, nan, seed in CASES:
            frames = make_frames(shape, count, nan, seed)
            store[name + "/frames"] = frames
            prev, nxt = frames[0].astype(np.float64), frames[1].astype(np.float64)
            pm, nm = np.ma.masked_invalid(prev), np.ma.masked_invalid(nxt)  # motion/lucaskanade.py:213-219
            np.ma.set_fill_value(pm, pm.min())
            np.ma.set_fill_value(nm, nm.min())
            opened_prev = images.morph_opening(pm.copy(), pm.min(), 3)  # utils/images.py:27-86
            opened_next = images.morph_opening(nm.copy(), nm.min(), 3)
            store[name + "/opened_prev"] = np.ma.filled(opened_prev, np.nan)
            store[name + "/opened_next"] = np.ma.filled(opened_next, np.nan)
            recorded.clear()
            pts = shitomasi.detection(opened_prev.copy(), max_corners=1000, max_num_features=None, quality_level=0.01,
                                      min_distance=10, block_size=5, buffer_mask=5, use_harris=False, k=0.04, verbose=False)
            store[name + "/corners"] = np.asarray(pts, dtype=np.float32)
            store[name + "/feature_u8"] = recorded["feature_u8"]
            if recorded.get("feature_mask") is not None:
                store[name + "/feature_mask"] = recorded["feature_mask"]
            xy, uv = tracking.track_features(opened_prev.copy(), opened_next.copy(), pts, winsize=(50, 50), nr_levels=3,
                                             criteria=(3, 10, 0), flags=0, min_eig_thr=1e-4, verbose=False)
            store[name + "/track_xy"], store[name + "/track_uv"] = np.asarray(xy), np.asarray(uv)
            store[name + "/prev_u8"], store[name + "/next_u8"] = recorded["prev_u8"], recorded["next_u8"]
            store[name + "/lk_p1"], store[name + "/lk_status"] = recorded["lk_p1"], recorded["lk_status"]
            sxy, suv = dense_lucaskanade(frames.astype(np.float64), dense=False)  # motion/lucaskanade.py:38-279
            store[name + "/sparse_xy"], store[name + "/sparse_uv"] = np.asarray(sxy), np.asarray(suv)
            store[name + "/dense"] = dense_lucaskanade(frames.astype(np.float64))
            print(name, "corners", len(pts), "tracked", len(xy), "sparse", len(sxy))
    finally:
        cv2.goodFeaturesToTrack, cv2.calcOpticalFlowPyrLK = real_gftt, real_lk
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
