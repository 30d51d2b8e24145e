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


# ---------------------------------------------------------------------------------------------
# SciPy twins: an independent second implementation of the separable pieces of the OpenCV
# restatement (cv2 is absent everywhere, so this does not pin the oracle against OpenCV; it removes
# "one author, one reading" from the morphology, the pyramid, the derivative stencils and the box
# sums - scipy.ndimage's own border handling and correlation code stand in for the hand-written
# index arithmetic of oracle/lk_opencv.py).  scipy mode "mirror" = BORDER_REFLECT_101.
# ---------------------------------------------------------------------------------------------
from scipy import ndimage as ndi

_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)


@pytest.mark.parametrize("shape,seed", [((37, 53), 0), ((64, 64), 1), ((5, 9), 2), ((120, 17), 3)])
def test_morph_opening_equals_scipy_binary_opening(shape, seed):
    """utils/images.py:58-86 through cv2.morphologyEx(OPEN, 3x3 cross): erosion with the border neutral
    (border_value=1), dilation with the border neutral (border_value=0)."""
    rng = np.random.default_rng(seed)
    img = np.where(rng.random(shape) < 0.45, rng.random(shape) * 30.0, -15.0).astype(np.float32)
    img[rng.random(shape) < 0.05] = np.nan
    valid = np.isfinite(img)
    fill = img[valid].min()
    got = lk.morph_opening(img, valid, fill)
    field = np.where(valid, img, fill) > fill
    eroded = ndi.binary_erosion(field, structure=_CROSS, border_value=1)
    opened = ndi.binary_dilation(eroded, structure=_CROSS, border_value=0)
    want = img.copy()
    want[field & ~opened] = fill
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(got[valid], want[valid])


@pytest.mark.parametrize("size", [1, 2, 3, 5, 6])
def test_dilate_mask_equals_scipy_binary_dilation(size):
    """cv2.dilate(mask, ones((size, size))) (shitomasi.py:137), anchor at size // 2: out(y, x) is the maximum over
    mask(y + i - size // 2, x + j - size // 2), 0 <= i, j < size - for an even size the window reaches one pixel
    further up / left than down / right.  That is scipy's maximum_filter with its default centre, and
    binary_dilation (which mirrors the structure) with origin -1 for even sizes."""
    rng = np.random.default_rng(size)
    mask = rng.random((40, 33)) < 0.03
    got = lk.dilate_mask(mask, size)
    want = ndi.maximum_filter(mask.astype(np.uint8), footprint=np.ones((size, size)), mode="constant") > 0
    assert np.array_equal(got, want)
    want2 = ndi.binary_dilation(mask, structure=np.ones((size, size), bool), origin=0 if size % 2 else -1)
    assert np.array_equal(got, want2)
    one = np.zeros((9, 9), bool)
    one[4, 4] = True
    ys, xs = np.nonzero(lk.dilate_mask(one, size))  # the pixels whose window contains (4, 4)
    assert ys.min() == 4 - (size - 1 - size // 2) and ys.max() == 4 + size // 2 and xs.min() == ys.min()


@pytest.mark.parametrize("shape", [(51, 64), (40, 40), (7, 9), (2, 33)])
def test_pyr_down_equals_scipy_correlation(shape):
    """cv::pyrDown 8U: [1 4 6 4 1] x [1 4 6 4 1], BORDER_REFLECT_101, (sum + 128) >> 8, every second pixel."""
    rng = np.random.default_rng(shape[0])
    u8 = rng.integers(0, 256, shape, dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1], np.int64)
    full = ndi.correlate1d(ndi.correlate1d(u8.astype(np.int64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    want = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)
    assert np.array_equal(lk.pyr_down(u8), want)


@pytest.mark.parametrize("shape", [(30, 41), (64, 64), (3, 5)])
def test_scharr_equals_scipy_correlation(shape):
    """calcSharrDeriv: Ix = [3 10 3]^T x [-1 0 1], Iy = [-1 0 1]^T x [3 10 3], BORDER_REFLECT_101, int16."""
    rng = np.random.default_rng(shape[1])
    u8 = rng.integers(0, 256, shape, dtype=np.uint8)
    a = u8.astype(np.int64)
    smooth, diff = np.array([3, 10, 3]), np.array([-1, 0, 1])
    want_x = ndi.correlate1d(ndi.correlate1d(a, smooth, axis=0, mode="mirror"), diff, axis=1, mode="mirror")
    want_y = ndi.correlate1d(ndi.correlate1d(a, diff, axis=0, mode="mirror"), smooth, axis=1, mode="mirror")
    ix, iy = lk.scharr_deriv(u8)
    assert np.array_equal(ix, want_x.astype(np.int16)) and np.array_equal(iy, want_y.astype(np.int16))


@pytest.mark.parametrize("block_size", [3, 5, 7])
def test_corner_min_eigenval_equals_scipy_sobel_and_box(block_size):
    """cornerMinEigenVal(8U, blockSize, ksize 3): Sobel derivatives scaled by 1 / (4 blockSize 255), products
    box-summed over blockSize x blockSize (BORDER_REFLECT_101 on the products), smaller eigenvalue.  The twin
    works in float64 throughout (scipy.ndimage.sobel / uniform_filter): agreement to float32 rounding of the
    response - the restatement's float32 roundings are what the device has to reproduce, the twin checks the
    STENCILS, scales and borders."""
    rng = np.random.default_rng(block_size)
    g = gaussian_filter(rng.standard_normal((48, 61)), 2.0)
    u8 = ((g - g.min()) / (g.max() - g.min()) * 255).astype(np.uint8)
    s = 1.0 / (4.0 * block_size * 255.0)
    a = u8.astype(np.float64)
    dx = ndi.sobel(a, axis=1, mode="mirror") * s
    dy = ndi.sobel(a, axis=0, mode="mirror") * s
    area = block_size * block_size
    box = lambda q: ndi.uniform_filter(q, size=block_size, mode="mirror") * area  # noqa: E731
    cxx, cxy, cyy = box(dx * dx), box(dx * dy), box(dy * dy)
    want = 0.5 * (cxx + cyy) - np.sqrt((0.5 * (cxx - cyy)) ** 2 + cxy * cxy)
    got = lk.corner_min_eigenval(u8, block_size).astype(np.float64)
    scale = np.abs(want).max()
    assert scale > 0
    assert np.abs(got - want).max() <= 2e-6 * scale


def test_reflect101_equals_numpy_pad_reflect():
    """_pad_reflect101 against numpy.pad(mode="reflect") (the same border rule, numpy's implementation)."""
    rng = np.random.default_rng(4)
    for shape, r in (((5, 7), 3), ((2, 9), 1), ((30, 30), 4)):
        a = rng.integers(0, 255, shape)
        if r < min(shape):
            assert np.array_equal(lk._pad_reflect101(a, r), np.pad(a, r, mode="reflect"))


def _lk_float_twin(prev_u8, next_u8, points, win=(50, 50), max_level=3, max_count=10):
    """A second, independent reading of pyramidal Lucas-Kanade (Bouguet 2001, the algorithm behind
    cv::calcOpticalFlowPyrLK), in float64 with SciPy's samplers: pyramid levels by correlate1d(mode="mirror") with
    OpenCV's rounding to uint8, Scharr derivative images (zero outside the image), window samples at
    pt - (win - 1) / 2 + (0 .. win - 1) by map_coordinates(order=1), the 2 x 2 normal equations solved per iteration,
    the oscillation stop of the OpenCV loop.  No fixed-point arithmetic, no shared code with oracle/lk_opencv.py."""
    from scipy import ndimage

    def down(img):
        k = np.array([1, 4, 6, 4, 1], dtype=np.int64)
        t = ndimage.correlate1d(img.astype(np.int64), k, axis=1, mode="mirror")
        t = ndimage.correlate1d(t, k, axis=0, mode="mirror")
        return ((t + 128) >> 8)[::2, ::2].astype(np.uint8)

    def levels(img):
        out = [img]
        for _ in range(max_level):
            nxt = down(out[-1])
            if nxt.shape[1] <= win[0] or nxt.shape[0] <= win[1]:
                break
            out.append(nxt)
        return out

    pi, pj = levels(prev_u8), levels(next_u8)
    top = min(len(pi), len(pj)) - 1
    w, h = win
    oy, ox = np.mgrid[0:h, 0:w].astype(np.float64)
    pts = np.asarray(points, dtype=np.float64)
    guess = np.zeros_like(pts)
    for level in range(top, -1, -1):
        I, J = pi[level].astype(np.float64), pj[level].astype(np.float64)
        smooth, diff = np.array([3.0, 10.0, 3.0]), np.array([-1.0, 0.0, 1.0])
        # (the Scharr stencil sums to 32 times the derivative)
        ix = ndimage.correlate1d(ndimage.correlate1d(I, diff, axis=1, mode="mirror"), smooth, axis=0, mode="mirror") / 32.0
        iy = ndimage.correlate1d(ndimage.correlate1d(I, diff, axis=0, mode="mirror"), smooth, axis=1, mode="mirror") / 32.0
        for i, p in enumerate(pts):
            u = p / (1 << level)
            ys, xs = u[1] - (h - 1) * 0.5 + oy, u[0] - (w - 1) * 0.5 + ox
            tpl = ndimage.map_coordinates(I, [ys, xs], order=1, mode="mirror")
            gx = ndimage.map_coordinates(ix, [ys, xs], order=1, mode="constant", cval=0.0)
            gy = ndimage.map_coordinates(iy, [ys, xs], order=1, mode="constant", cval=0.0)
            G = np.array([[np.sum(gx * gx), np.sum(gx * gy)], [np.sum(gx * gy), np.sum(gy * gy)]])
            d = guess[i] * 2.0 if level != top else np.zeros(2)
            prev = np.zeros(2)
            for it in range(max_count):
                cur = ndimage.map_coordinates(J, [ys + d[1], xs + d[0]], order=1, mode="mirror")
                err = cur - tpl
                b = -np.array([np.sum(err * gx), np.sum(err * gy)])
                step = np.linalg.solve(G, b)
                d = d + step
                if it > 0 and np.all(np.abs(step + prev) < 0.01):
                    d = d - 0.5 * step
                    break
                prev = step
            guess[i] = d
    return pts + guess


@pytest.mark.parametrize("shift,seed", [((1.3, -0.7), 0), ((-2.6, 3.2), 1), ((0.25, 0.4), 2)])
def test_tracker_restatement_tracks_like_an_independent_float_lucas_kanade(shift, seed):
    """The OpenCV restatement's tracker (fixed-point patches, 2^-20 scaling, int64 sums) against a float64 pyramidal
    Lucas-Kanade written from the algorithm's description with SciPy's samplers: the two agree to 1e-4 px on smooth
    textures - what the fixed-point arithmetic of the original costs - and both recover the sub-pixel shift.  Not a pin against OpenCV: a second author for row a8."""
    from scipy import ndimage

    rng = np.random.default_rng(seed)
    m, n = 160, 176
    base = ndimage.gaussian_filter(rng.normal(size=(m + 40, n + 40)), 3.0)
    base = (base - base.min()) / (base.max() - base.min()) * 255.0
    yy, xx = np.mgrid[0:m, 0:n].astype(np.float64) + 20.0
    prev = ndimage.map_coordinates(base, [yy, xx], order=3)
    nxt = ndimage.map_coordinates(base, [yy - shift[1], xx - shift[0]], order=3)
    a8, b8 = np.clip(prev, 0, 255).astype(np.uint8), np.clip(nxt, 0, 255).astype(np.uint8)
    pts = np.array([[60.0, 70.0], [88.5, 61.25], [100.0, 90.0], [70.75, 99.5]], dtype=np.float32)
    got, status = lk.calc_optical_flow_pyr_lk(a8, b8, pts, win=(31, 31), max_level=2)
    twin = _lk_float_twin(a8, b8, pts, win=(31, 31), max_level=2)
    assert status.all()
    assert np.max(np.abs(got - twin)) < 1e-3, (got - twin)  # observed: 0.9e-4 ... 1.2e-4 px
    assert np.max(np.abs((got - pts) - np.array(shift))) < 0.03  # observed: <= 0.012 px


@pytest.mark.parametrize("min_distance,max_corners,seed", [(10, 1000, 0), (4, 60, 1), (7, 0, 2)])
def test_corner_selection_equals_a_brute_force_reading(min_distance, max_corners, seed):
    """goodFeaturesToTrack's selection stage read a second time, without the cell grid: threshold at quality x max,
    3 x 3 non-maximum suppression by scipy.ndimage.maximum_filter, one-pixel border excluded, candidates in descending
    (value, address) order, each accepted unless an ALREADY ACCEPTED corner lies closer than min_distance (all pairs
    compared).  For integral distances the grid of the original finds exactly these corners, in this order."""
    from scipy import ndimage

    rng = np.random.default_rng(seed)
    img = gaussian_filter(rng.normal(size=(90, 120)), 2.0)
    u8 = ((img - img.min()) / (img.max() - img.min()) * 255).astype(np.uint8)
    allowed = np.ones(u8.shape, dtype=bool)
    allowed[:, 100:] = False
    got = lk.good_features_to_track(u8, allowed, max_corners=max_corners, quality=0.01, min_distance=min_distance, block_size=5)

    eig = lk.corner_min_eigenval(u8, 5)
    thr = np.float32(eig[allowed].max() * 0.01)
    kept = np.where(eig > thr, eig, np.float32(0))
    peak = ndimage.maximum_filter(kept, size=3, mode="constant", cval=-np.inf)
    cand = (kept != 0) & (kept == peak) & allowed
    cand[[0, -1], :] = False
    cand[:, [0, -1]] = False
    ys, xs = np.nonzero(cand)
    order = sorted(range(len(ys)), key=lambda i: (-float(kept[ys[i], xs[i]]), -(int(ys[i]) * u8.shape[1] + int(xs[i]))))
    accepted = []
    for i in order:
        x, y = int(xs[i]), int(ys[i])
        if all((x - ax) ** 2 + (y - ay) ** 2 >= min_distance ** 2 for ax, ay in accepted):
            accepted.append((x, y))
            if 0 < max_corners == len(accepted):
                break
    want = np.array(accepted, dtype=np.float32).reshape(-1, 2)
    assert len(want) > 5
    np.testing.assert_array_equal(got, want)


def test_reference_orchestration_around_a_standin_cv2_gives_the_oracle_pipeline(ref_pysteps):
    """The REAL pysteps.motion.lucaskanade.dense_lucaskanade, run in a process of its own around a stand-in ``cv2``
    whose five functions are the restated OpenCV algorithms: the sparse vectors are the oracle pipeline's bit for bit
    and the dense fields identical - rows a4 / a5 (NaN masking, minimum fill, opening field, uint8 renderings, buffer
    mask with its row quirk, the cv2 call conventions, pooling, outlier test, declustering, interpolation) held
    against the reference's own code; every one of the five call sites is exercised (tests/helpers/)."""
    import json
    import os
    import subprocess
    import sys

    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "ref_lk_with_standin_cv2.py")
    run = subprocess.run([sys.executable, helper], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    report = json.loads(run.stdout.strip().splitlines()[-1])
    assert all(count > 0 for count in report["calls"].values()), report["calls"]
    for case in ("plain", "nan_three_frames"):
        r = report[case]
        assert r["vectors"] > 50 and r["sparse_equal"] and r["dense_max_abs_diff"] == 0.0, r
