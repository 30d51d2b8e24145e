"""Property pins of the OpenCV restatement (oracle/lk_opencv.py) - no cv2, no GPU.

No golden vectors exist for the OpenCV stages (parity unpinned, see the oracle's
header); these are the reference's own property tests restated on data-free inputs:
  pysteps/tests/test_motion.py:154-250   uniform shift recovered (rel. RMSE)
  pysteps/tests/test_motion.py:265-289   all-zero input -> |uv| < 0.01
  pysteps/tests/test_motion_lk.py:86-105 output formats; nr_std_outlier=0 -> zero field
plus known answers of the building blocks that can be derived by hand.
"""

import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

from oracle import lk_opencv as lk


def _texture(m, n, seed=0, sigma=3.0):
    rng = np.random.default_rng(seed)
    g = gaussian_filter(rng.standard_normal((m, n)), sigma, mode="wrap")
    return ((g - g.min()) / (g.max() - g.min()) * 40.0 - 15.0).astype(np.float32)


def test_structuring_element_is_cross():
    assert np.array_equal(lk._structuring_element_ellipse(3), [[0, 1, 0], [1, 1, 1], [0, 1, 0]])


def test_opening_removes_speckle_keeps_blobs():
    img = np.full((12, 12), -15.0, dtype=np.float32)
    img[2, 2] = 5.0                      # isolated pixel: removed
    img[6:9, 6:9] = 7.0                  # 3x3 block: centre cross survives, corners go
    img[5, 0] = 3.0                      # touches the border: neutral border cannot save it alone
    out = lk.morph_opening(img, np.ones_like(img, bool), img.min())
    assert out[2, 2] == -15.0 and out[5, 0] == -15.0
    assert out[7, 7] == 7.0 and out[6, 7] == 7.0 and out[7, 6] == 7.0
    assert out[6, 6] == -15.0 and out[8, 8] == -15.0


def test_pyrdown_constant_and_size():
    a = np.full((51, 64), 37, dtype=np.uint8)
    b = lk.pyr_down(a)
    assert b.shape == (26, 32) and np.all(b == 37)


def test_scharr_of_ramp():
    ramp = np.tile(np.arange(40, dtype=np.uint8) * 3, (20, 1))
    ix, iy = lk.scharr_deriv(ramp)
    assert np.all(ix[:, 1:-1] == 3 * 2 * 16) and np.all(iy == 0)
    assert np.all(ix[:, 0] == 0)  # reflect-101: symmetric neighbours at the edge


def test_corner_response_flat_and_corner():
    flat = np.full((32, 32), 100, dtype=np.uint8)
    assert np.all(lk.corner_min_eigenval(flat) == 0)
    img = np.zeros((40, 40), dtype=np.uint8)
    img[20:, 20:] = 200
    pts = lk.good_features_to_track(img, np.ones(img.shape, bool))
    assert pts.shape[0] >= 1 and np.all(np.abs(pts[0] - [20, 20]) <= 2)


def test_min_distance_and_max_corners():
    tex = _texture(128, 128)
    u8 = lk.to_uint8(tex, np.ones(tex.shape, bool), tex.min(), tex.max(), tex.min())
    pts = lk.good_features_to_track(u8, np.ones(u8.shape, bool), max_corners=15, min_distance=10)
    assert 0 < pts.shape[0] <= 15
    d = np.hypot(*(pts[:, None, :] - pts[None, :, :]).transpose(2, 0, 1))
    assert d[~np.eye(len(pts), dtype=bool)].min() >= 10


@pytest.mark.parametrize("shift", [(2, 0), (0, 2), (3, -2)])
def test_tracker_recovers_exact_shift(shift):
    tex = _texture(192, 192, seed=1)
    a = lk.to_uint8(tex, np.ones(tex.shape, bool), tex.min(), tex.max(), tex.min())
    b = np.roll(a, (shift[1], shift[0]), axis=(0, 1))
    pts = lk.good_features_to_track(a, np.ones(a.shape, bool))
    inner = np.all((pts > 50) & (pts < 142), axis=1)
    p1, st = lk.calc_optical_flow_pyr_lk(a, b, pts[inner])
    assert st.all()
    assert np.abs(p1 - pts[inner] - shift).max() < 5e-3


def test_dense_uniform_shift_rel_rmse():
    tex = _texture(160, 160, seed=2)
    frames = np.stack([np.roll(tex, (0, 2 * t), axis=(0, 1)) for t in range(3)])
    field = lk.dense_lucaskanade(frames)
    assert field.shape == (2, 160, 160) and field.dtype == np.float64
    inner = (slice(None), slice(40, 120), slice(40, 120))
    ideal = np.zeros_like(field)
    ideal[0] = 2.0
    rel_rmse = np.sqrt(((ideal - field)[inner] ** 2).mean() / (ideal[inner] ** 2).mean()) * 100
    # the reference's threshold is 0.1 % on a radar composite shifted with vet.morph
    # (test_motion.py:154-158); this periodic synthetic texture has border features
    # whose windows see reflected instead of wrapped data
    assert rel_rmse < 0.5


def test_zeros_give_zero_motion_and_formats():
    z = lk.dense_lucaskanade(np.zeros((2, 100, 100), np.float32))
    assert z.shape == (2, 100, 100) and np.abs(z).max() < 0.01
    xy, uv = lk.dense_lucaskanade(np.zeros((2, 100, 100), np.float32), dense=False)
    assert xy.shape == (0, 2) and uv.shape == (0, 2)
    tex = _texture(128, 128, seed=4)
    frames = np.stack([tex, np.roll(tex, 1, axis=1)])
    assert np.all(lk.dense_lucaskanade(frames, nr_std_outlier=0) == 0)
    xy, uv = lk.dense_lucaskanade(frames, dense=False)
    assert xy.ndim == 2 and xy.shape[1] == 2 and uv.shape == xy.shape


@pytest.mark.parametrize("min_distance,max_corners", [(10, 1000), (3.5, 200), (0.5, 50), (25, 40)])
def test_native_greedy_pass_matches_oracle(min_distance, max_corners):
    """psh_lk_greedy_host (the host half of psh_lk_corners_*: ordered min-distance acceptance on a
    cell grid, pure C++) walks the candidates exactly like the oracle's goodFeaturesToTrack."""
    import ctypes

    from pysteps_amd import _lib

    lib = _lib.load()  # dlopen only: no GPU needed
    rng = np.random.default_rng(int(min_distance * 10) + max_corners)
    m, n = 180, 240
    from scipy.ndimage import gaussian_filter

    img = gaussian_filter(rng.standard_normal((m, n)), 1.5)
    u8 = np.clip((img - img.min()) / (img.max() - img.min()) * 255, 0, 255).astype(np.uint8)
    allowed = np.ones((m, n), bool)
    want = lk.good_features_to_track(u8, allowed, max_corners=max_corners, min_distance=min_distance)
    # the candidate list the device hands over: thresholded 3x3 maxima, strongest first,
    # ties by higher address first (response bits << 32 | address, descending)
    eig = lk.corner_min_eigenval(u8, 5)
    thr = np.float32(eig.max() * 0.01)
    e = np.where(eig > thr, eig, np.float32(0))
    pad = np.full((m + 2, n + 2), -np.inf, dtype=np.float32)
    pad[1:-1, 1:-1] = e
    dil = np.max([pad[i:i + m, j:j + n] for i in range(3) for j in range(3)], axis=0)
    cand = (e != 0) & (e == dil)
    cand[0, :] = cand[-1, :] = False
    cand[:, 0] = cand[:, -1] = False
    ys, xs = np.nonzero(cand)
    keys = (e[ys, xs].view(np.uint32).astype(np.uint64) << np.uint64(32)) | (ys * n + xs).astype(np.uint64)
    keys = np.sort(keys)[::-1].copy()
    pts = np.empty((max_corners, 2), dtype=np.float32)
    count = ctypes.c_int(0)
    rc = lib.psh_lk_greedy_host(keys.ctypes.data, int(keys.size), m, n, float(min_distance), int(max_corners),
                                pts.ctypes.data, ctypes.byref(count))
    assert rc == 0
    assert count.value == len(want)
    assert np.array_equal(pts[: count.value], want)
    # argument errors do not touch the device either
    assert lib.psh_lk_greedy_host(keys.ctypes.data, -1, m, n, 10.0, 10, pts.ctypes.data, ctypes.byref(count)) != 0
