"""HIP dense Lucas-Kanade vs the OpenCV restatement (oracle/lk_opencv.py).

LK parity is UNPINNED at the OpenCV boundary (no cv2 here, no golden vectors in
the reference): the bar is (i) HIP vs the repo's CPU restatement - integer stages
bit-exact, sparse vectors within 1e-2 px, dense field rel-L2 <= 1e-3 - and (ii) the
reference's property tests (pysteps/tests/test_motion.py:154-289,400-430,
test_motion_lk.py:86-105) on data-free inputs.  Everything goes through the C ABI.
"""

import numpy as np
import pytest
from scipy.ndimage import gaussian_filter

from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _texture(m, n, seed=0, sigma=3.0):
    rng = np.random.default_rng(seed)
    g = gaussian_filter(rng.standard_normal((m, n)), sigma, mode="wrap")
    return ((g - g.min()) / (g.max() - g.min()) * 40.0 - 15.0).astype(np.float32)


def _rain(m, n, seed):
    from tools import synth

    return synth.rain_field_db(m, n, seed=seed, sigma=max(m / 64.0, 2.0))


@pytest.fixture(scope="module")
def lkmod():
    from pysteps_amd.motion import lucaskanade

    return lucaskanade


@pytest.fixture(scope="module")
def dense_lk():
    from pysteps_amd.motion import get_method

    return get_method("LK")


def _prep(lkmod, frame, size_opening=3, buffer_mask=5):
    from pysteps_amd.device import DeviceArray

    return lkmod.PreparedFrame(DeviceArray.from_host(frame, dtype=np.float32), size_opening, buffer_mask, True)


@pytest.mark.parametrize("shape,nan", [((96, 130), False), ((257, 64), True), ((5, 7), False), ((300, 300), True)])
def test_prepare_bit_exact(lkmod, shape, nan):
    """opening + both uint8 renderings are integer/byte work: bit-exact vs the oracle."""
    from oracle import lk_opencv as olk

    m, n = shape
    img = _rain(m, n, seed=m)
    rng = np.random.default_rng(n)
    speck = rng.random((m, n)) < 0.02
    img[speck] = rng.uniform(0, 30, speck.sum()).astype(np.float32)  # isolated pixels for the opening
    if nan:
        img[: m // 5, : n // 3] = np.nan
        img[m // 2, n // 2] = np.nan
    valid = np.isfinite(img)
    prep = _prep(lkmod, img)
    clean = olk.morph_opening(img, valid, img[valid].min())
    got_clean = prep.clean.to_host()
    assert np.array_equal(np.isnan(got_clean), ~valid)
    assert np.array_equal(got_clean[valid], clean[valid])
    lo, hi = clean[valid].min(), clean[valid].max()
    assert np.array_equal(prep.track_u8.to_host(), olk.to_uint8(clean, valid, lo, hi, lo))
    use = valid.copy()
    use[0, :] = False
    if (~valid).any() and m > 1:
        use[1, :] = False
    if use.any():
        flo, fhi = clean[use].min(), clean[use].max()
        assert np.array_equal(prep.feature_u8.to_host(), olk.to_uint8(clean, use, flo, fhi, lo))


@pytest.mark.parametrize("shape,nan", [((128, 160), False), ((200, 200), True), ((512, 384), False)])
def test_corners_match_oracle(lkmod, shape, nan):
    from oracle import lk_opencv as olk

    m, n = shape
    img = _rain(m, n, seed=7 + m)
    if nan:
        img[:30, :50] = np.nan
        img[100:110, 120:140] = np.nan
    valid = np.isfinite(img)
    clean = olk.morph_opening(img, valid, img[valid].min())
    want = olk.shitomasi_detection(clean, valid)
    got = lkmod.detect_corners(_prep(lkmod, img))
    assert got.dtype == np.float32 and got.shape[1] == 2 and got.shape[0] > 0
    # corner selection is index work (response ordering + greedy min-distance pass): the same
    # corners in the same order, no slack
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shape,shift", [((192, 192), (2, -1)), ((300, 260), (-3, 2)), ((130, 520), (1, 1))])
def test_tracker_matches_oracle(lkmod, shape, shift):
    from oracle import lk_opencv as olk

    m, n = shape
    a = _texture(m, n, seed=m)
    b = np.roll(a, (shift[1], shift[0]), axis=(0, 1)) + _texture(m, n, seed=n + 1) * 0.02
    pa, pb = _prep(lkmod, a, 0, 0), _prep(lkmod, b, 0, 0)
    ones = np.ones((m, n), bool)
    a8 = olk.to_uint8(a, ones, a.min(), a.max(), a.min())
    b8 = olk.to_uint8(b, ones, b.min(), b.max(), b.min())
    assert np.array_equal(pa.track_u8.to_host(), a8) and np.array_equal(pb.track_u8.to_host(), b8)
    pts = olk.good_features_to_track(a8, ones)
    extra = np.array([[0.0, 0.0], [n - 1.0, m - 1.0], [3.5, m / 2.0], [n / 2.0, 1.25]], dtype=np.float32)
    pts = np.vstack([pts, extra]).astype(np.float32)
    want, wst = olk.calc_optical_flow_pyr_lk(a8, b8, pts)
    got, gst = lkmod.track_points(pa, pb, pts)
    assert np.array_equal(gst, wst)
    assert np.abs(got[wst] - want[wst]).max() < 1e-2
    inner = np.all((pts > 60) & (pts < np.array([n, m]) - 60), axis=1) & wst
    if inner.any():
        assert np.abs((got - pts)[inner] - shift).max() < 0.05


@pytest.mark.parametrize("win", [(21, 21), (15, 31), (63, 40), (64, 64), (37, 64)])
def test_tracker_window_sizes(lkmod, win):
    """Every tracker instantiation (row-structured 8 / 13 / 16 rows per wave, and the gather kernel
    that takes 64-column windows) against the oracle."""
    from oracle import lk_opencv as olk

    m, n = 260, 300
    a = _texture(m, n, seed=win[0])
    b = np.roll(a, (1, -2), axis=(0, 1)) + _texture(m, n, seed=win[1] + 7) * 0.02
    pa, pb = _prep(lkmod, a, 0, 0), _prep(lkmod, b, 0, 0)
    ones = np.ones((m, n), bool)
    a8 = olk.to_uint8(a, ones, a.min(), a.max(), a.min())
    b8 = olk.to_uint8(b, ones, b.min(), b.max(), b.min())
    pts = olk.good_features_to_track(a8, ones, max_corners=150)
    pts = np.vstack([pts, [[0.0, 0.0], [n - 1.0, m - 1.0], [2.5, m / 2.0]]]).astype(np.float32)
    want, wst = olk.calc_optical_flow_pyr_lk(a8, b8, pts, win=win)
    got, gst = lkmod.track_points(pa, pb, pts, winsize=win)
    assert np.array_equal(gst, wst)
    assert np.abs(got[wst] - want[wst]).max() < 1e-2


def test_tracker_small_image_fewer_levels(lkmod):
    """levels are dropped until the image is larger than the window (buildOpticalFlowPyramid)."""
    from oracle import lk_opencv as olk

    a = _texture(100, 120, seed=3)
    b = np.roll(a, 1, axis=1)
    pa, pb = _prep(lkmod, a, 0, 0), _prep(lkmod, b, 0, 0)
    ones = np.ones(a.shape, bool)
    a8 = olk.to_uint8(a, ones, a.min(), a.max(), a.min())
    b8 = olk.to_uint8(b, ones, b.min(), b.max(), b.min())
    pts = olk.good_features_to_track(a8, ones)
    want, wst = olk.calc_optical_flow_pyr_lk(a8, b8, pts)
    got, gst = lkmod.track_points(pa, pb, pts)
    assert np.array_equal(gst, wst) and np.abs(got[wst] - want[wst]).max() < 1e-2


# ---- end to end ---------------------------------------------------------------
def _advected_frames(m, n, count, seed, nan=False):
    from oracle import semilag_cport as ocl
    from tools import synth

    base = synth.rain_field_db(m, n, seed=seed, sigma=max(m / 96.0, 2.0))
    vel = synth.true_velocity(m, n)
    adv = ocl.extrapolate(base, vel, count - 1, outval=-15.0)
    frames = np.stack([base] + [adv[t] for t in range(count - 1)])
    if nan:
        frames[:, synth.border_nan_mask(m, n, 0.1)] = np.nan
    return frames, vel


@pytest.mark.parametrize("m,n,count,nan", [(256, 256, 2, False), (384, 320, 3, False), (320, 320, 2, True)])
def test_dense_lk_matches_oracle(dense_lk, m, n, count, nan):
    from oracle import lk_opencv as olk

    frames, vel = _advected_frames(m, n, count, seed=m + count, nan=nan)
    wxy, wuv = olk.dense_lucaskanade(frames, dense=False)
    gxy, guv = dense_lk(frames, dense=False)
    assert gxy.dtype == np.float64 and gxy.shape[1] == 2 and guv.shape == gxy.shape
    assert abs(len(gxy) - len(wxy)) <= max(3, 0.03 * len(wxy))
    wmap = {}
    for p, v in zip(wxy.astype(int), wuv):  # the same pixel can be a feature in several frame pairs
        wmap.setdefault(tuple(p), []).append(v)
    d = [min(np.abs(v - w).max() for w in wmap[tuple(p)])
         for p, v in zip(gxy.astype(int), guv) if tuple(p) in wmap]
    assert len(d) >= 0.95 * len(wxy)
    assert max(d) < 1e-2
    want = olk.dense_lucaskanade(frames)
    got = dense_lk(frames)
    assert got.shape == (2, m, n) and got.dtype == np.float64
    assert rel_l2(got, want) < 1e-3
    # and the motion it was made with is recovered (away from the inflow border)
    inner = (slice(None), slice(m // 4, 3 * m // 4), slice(n // 4, 3 * n // 4))
    assert np.sqrt(np.mean((got - vel)[inner] ** 2)) < 0.5


@pytest.mark.parametrize("size,count,nr_levels", [(1024, 2, 3), (1024, 2, 2), (2048, 3, 3), (2048, 3, 2), (4096, 2, 3)])
def test_lk_parity_at_baseline_sizes(lkmod, dense_lk, size, count, nr_levels):
    """LK against the restatement where the 1000-corner cap, the 4-level pyramid and the
    min-distance grid bind (>= 1024^2; BASELINE config 2 = 2048^2, 3 frames; config 3, the headline
    = 4096^2, 2 frames): the reference default
    ``nr_levels=3`` (maxLevel 3 = 4 pyramid levels) and config 2's "3-level" pyramid
    (``nr_levels=2``, SURVEY 8d "run both").  Integer stages bit-exact (opening, uint8 renderings,
    corner list incl. order, tracking status), vectors <= 1e-2 px, dense field <= 1e-3 rel-L2 on a
    pixel lattice (the float64 k-d tree IDW of the oracle on 4M pixels takes half a minute)."""
    from oracle import lk_opencv as olk
    from oracle import sparse as osp

    m = n = size
    frames, vel = _advected_frames(m, n, count, seed=size + nr_levels)
    lk_kwargs = {"nr_levels": nr_levels}
    # --- stage by stage on the first pair ------------------------------------------------
    valid = np.ones((m, n), bool)
    prev, nxt = frames[0], frames[1]
    cprev = olk.morph_opening(prev, valid, prev.min())
    cnxt = olk.morph_opening(nxt, valid, nxt.min())
    pa, pb = _prep(lkmod, prev), _prep(lkmod, nxt)
    assert np.array_equal(pa.clean.to_host(), cprev) and np.array_equal(pb.clean.to_host(), cnxt)
    a8 = olk.to_uint8(cprev, valid, cprev.min(), cprev.max(), cprev.min())
    b8 = olk.to_uint8(cnxt, valid, cnxt.min(), cnxt.max(), cnxt.min())
    assert np.array_equal(pa.track_u8.to_host(), a8) and np.array_equal(pb.track_u8.to_host(), b8)
    want_pts = olk.shitomasi_detection(cprev, valid)
    got_pts = lkmod.detect_corners(pa)
    assert len(want_pts) == 1000  # the cap binds at these sizes
    assert np.array_equal(got_pts, want_pts)
    want_p1, wst = olk.calc_optical_flow_pyr_lk(a8, b8, want_pts, max_level=nr_levels)
    got_p1, gst = lkmod.track_points(pa, pb, got_pts, nr_levels=nr_levels)
    assert np.array_equal(gst, wst)
    assert np.abs(got_p1[wst] - want_p1[wst]).max() < 1e-2
    # --- end to end: pooled, outlier-filtered vectors and the dense field -----------------
    wxy, wuv = olk.dense_lucaskanade(frames, dense=False, nr_levels=nr_levels)
    gxy, guv = dense_lk(frames, dense=False, lk_kwargs=lk_kwargs)
    assert np.array_equal(gxy, wxy)  # same features survive, same order
    assert np.abs(guv - wuv).max() < 1e-2
    got = dense_lk(frames, lk_kwargs=lk_kwargs)
    dxy, duv = osp.decluster(wxy, wuv, 20, 1)
    step = 5 if size <= 1024 else (9 if size <= 2048 else 17)
    ys, xs = np.arange(2, m, step), np.arange(3, n, step)
    want = _idw_lattice(dxy, duv, xs, ys)
    sub = got[:, ys[:, None], xs[None, :]]
    assert rel_l2(sub, want) < 1e-3
    inner = (slice(None), slice(m // 4, 3 * m // 4), slice(n // 4, 3 * n // 4))
    assert np.sqrt(np.mean((got - vel)[inner] ** 2)) < 0.3


def test_lk_parity_at_8192_config5(lkmod, dense_lk):
    """BASELINE config 5 (8192^2, "4-level pyramid" = ``nr_levels=3``) on ONE GPU, first frame pair,
    stage by stage against the restatement: opening and uint8 renderings bit-exact, the 1000-corner list
    identical incl. order (about 4x the candidates of 4096^2 in the selection chunks), tracking status
    equal, end points <= 1e-2 px; then the pooled vectors of the whole call.  The frames are made on the
    device (the extrapolator is only the input generator here; both sides see the same arrays)."""
    from oracle import lk_opencv as olk
    from pysteps_amd import extrapolation
    from tools import synth

    m = n = 8192
    base = synth.rain_field_db(m, n, seed=85, sigma=m / 96.0)
    vel = synth.true_velocity(m, n)
    adv = extrapolation.get_method("semilagrangian")(base, vel, 1, outval=-15.0)
    frames = np.stack([base, adv[0]])
    valid = np.ones((m, n), bool)
    cprev = olk.morph_opening(frames[0], valid, frames[0].min())
    cnxt = olk.morph_opening(frames[1], valid, frames[1].min())
    pa, pb = _prep(lkmod, frames[0]), _prep(lkmod, frames[1])
    assert np.array_equal(pa.clean.to_host(), cprev) and np.array_equal(pb.clean.to_host(), cnxt)
    a8 = olk.to_uint8(cprev, valid, cprev.min(), cprev.max(), cprev.min())
    b8 = olk.to_uint8(cnxt, valid, cnxt.min(), cnxt.max(), cnxt.min())
    assert np.array_equal(pa.track_u8.to_host(), a8) and np.array_equal(pb.track_u8.to_host(), b8)
    want_pts = olk.shitomasi_detection(cprev, valid)
    got_pts = lkmod.detect_corners(pa)
    assert len(want_pts) == 1000
    assert np.array_equal(got_pts, want_pts)
    want_p1, wst = olk.calc_optical_flow_pyr_lk(a8, b8, want_pts, max_level=3)
    got_p1, gst = lkmod.track_points(pa, pb, got_pts, nr_levels=3)
    assert np.array_equal(gst, wst)
    assert np.abs(got_p1[wst] - want_p1[wst]).max() < 1e-2
    # the pooled, outlier-filtered vectors of the whole call from the stages above (lucaskanade.py:203-260)
    from oracle import sparse as osp

    xy = want_pts[wst].astype(np.float64)
    uv = (want_p1[wst] - want_pts[wst]).astype(np.float64)
    keep = ~osp.detect_outliers(uv, 3, xy, 30)
    gxy, guv = dense_lk(frames, dense=False, lk_kwargs={"nr_levels": 3})
    assert np.array_equal(gxy, xy[keep])
    assert np.abs(guv - uv[keep]).max() < 1e-2
    got = dense_lk(frames, lk_kwargs={"nr_levels": 3})
    inner = (slice(None), slice(m // 4, 3 * m // 4), slice(n // 4, 3 * n // 4))
    assert np.sqrt(np.mean((got - vel)[inner] ** 2)) < 0.3


def _idw_lattice(xy, values, xs, ys, k=20, power=0.5, dist_offset=0.5):
    """interpolate.py:80-109 on the lattice xs x ys (float64 k-d tree, as oracle.sparse.idw)."""
    from scipy.spatial import cKDTree

    gx, gy = np.meshgrid(xs, ys)
    dist, inds = cKDTree(xy).query(np.column_stack([gx.ravel(), gy.ravel()]), k=min(k, len(xy)))
    w = 1.0 / np.power(dist + dist_offset, power)
    w /= w.sum(axis=1, keepdims=True)
    out = np.sum(values[inds, :] * w[..., None], axis=1)
    return np.moveaxis(out.reshape(len(ys), len(xs), 2), -1, 0)


def test_uniform_shift_recovered(dense_lk):
    tex = _texture(256, 256, seed=2)
    for axis, comp in ((2, 0), (1, 1)):
        frames = np.stack([np.roll(tex, 2 * t, axis=axis - 1) for t in range(3)])
        field = dense_lk(frames)
        ideal = np.zeros_like(field)
        ideal[comp] = 2.0
        inner = (slice(None), slice(64, 192), slice(64, 192))
        rel_rmse = np.sqrt(((ideal - field)[inner] ** 2).mean() / (ideal[inner] ** 2).mean()) * 100
        assert rel_rmse < 0.1  # the reference's own bar (pysteps/tests/test_motion.py:154-158); observed 0.03


def test_no_precipitation_and_formats(dense_lk):
    z = dense_lk(np.zeros((2, 100, 100)))
    assert z.shape == (2, 100, 100) and np.abs(z).max() < 0.01
    xy, uv = dense_lk(np.zeros((2, 100, 100)), dense=False)
    assert xy.shape == (0, 2) and uv.shape == (0, 2)
    for count in (2, 3, 5, 12):
        assert dense_lk(np.zeros((count, 100, 100))).shape == (2, 100, 100)
    with pytest.raises(ValueError):
        dense_lk(np.zeros(100))
    with pytest.raises(ValueError):
        dense_lk(np.zeros((100, 100)))
    tex = _texture(128, 128, seed=4)
    frames = np.stack([tex, np.roll(tex, 1, axis=1)])
    assert np.all(dense_lk(frames, nr_std_outlier=0) == 0)
    xy, uv = dense_lk(frames, dense=False)
    assert xy.ndim == 2 and xy.shape[1] == 2 and uv.shape == xy.shape and len(xy) > 0
    few = dense_lk(frames, fd_kwargs={"max_num_features": 15}, dense=False)[0]
    assert 0 < len(few) <= 15


def test_nan_equals_masked_input(dense_lk):
    """pysteps/tests/test_motion.py:400-430: NaN ndarray and MaskedArray give the same field."""
    frames, _ = _advected_frames(256, 256, 2, seed=11, nan=True)
    masked = np.ma.masked_invalid(frames)
    a = dense_lk(frames, fd_kwargs={"buffer_mask": 20})
    b = dense_lk(masked, fd_kwargs={"buffer_mask": 20})
    assert np.abs(a - b).max() < 0.01


def test_device_resident_frames(dense_lk):
    from pysteps_amd.device import DeviceArray

    frames, _ = _advected_frames(256, 256, 2, seed=5)
    host = dense_lk(frames)
    dev = dense_lk(DeviceArray.from_host(frames))
    assert isinstance(dev, DeviceArray) and dev.shape == (2, 256, 256) and dev.dtype == np.float32
    assert np.abs(dev.to_host() - host).max() < 1e-4


def test_dense_lk_4096_recovers_true_motion(dense_lk):
    """Bench workload (config 3) at full size, device resident: the frames are the base field
    advected by the known motion; dense LK must give that motion back away from the inflow edges."""
    from pysteps_amd import _lib, extrapolation
    from pysteps_amd.device import DeviceArray
    from tools import synth

    m = n = 4096
    base = synth.rain_field_db(m, n)
    vel = synth.true_velocity(m, n)
    vel_d = DeviceArray.from_host(vel)
    frames = DeviceArray((2, m, n), np.float32)
    lib = _lib.lib()
    _lib.check(lib.psh_memcpy_h2d(frames.ptr, base.ctypes.data, base.nbytes))
    adv = extrapolation.get_method("semilagrangian")(frames.view(0), vel_d, 1, outval=-15.0)
    _lib.check(lib.psh_memcpy_d2d(frames.view(1).ptr, adv.ptr, adv.nbytes))
    field = dense_lk(frames)
    assert isinstance(field, DeviceArray) and field.shape == (2, m, n)
    got = field.to_host()
    assert np.isfinite(got).all()
    inner = (slice(None), slice(512, m - 512), slice(512, n - 512))
    rmse = np.sqrt(np.mean((got - vel)[inner] ** 2))
    assert rmse < 0.25, rmse
    xy, uv = dense_lk(frames.to_host(), dense=False)
    assert 500 < len(xy) <= 1000


def test_native_and_python_orchestration_agree(lkmod, dense_lk):
    """psh_dense_lk_dev (one C call) and the stage-by-stage Python loop give the same results."""
    frames, _ = _advected_frames(320, 288, 3, seed=17, nan=True)
    try:
        lkmod.USE_NATIVE_ORCHESTRATION = False
        py_field = dense_lk(frames)
        py_xy, py_uv = dense_lk(frames, dense=False)
        py_opts = dense_lk(frames, fd_kwargs={"max_num_features": 40, "buffer_mask": 0}, decl_scale=0,
                           interp_kwargs={"k": 5, "power": 2.0})
    finally:
        lkmod.USE_NATIVE_ORCHESTRATION = True
    field = dense_lk(frames)
    xy, uv = dense_lk(frames, dense=False)
    opts = dense_lk(frames, fd_kwargs={"max_num_features": 40, "buffer_mask": 0}, decl_scale=0,
                    interp_kwargs={"k": 5, "power": 2.0})
    assert np.array_equal(xy, py_xy) and np.array_equal(uv, py_uv)
    assert np.array_equal(field, py_field)
    assert np.array_equal(opts, py_opts)


@pytest.mark.parametrize("shape", [(30, 30), (51, 64), (64, 200), (7, 300)])
def test_small_images_do_not_break(dense_lk, shape):
    """Images smaller than / comparable to the 50x50 tracking window: same behaviour as the
    restatement (features whose window cannot be placed are dropped), never a crash."""
    from oracle import lk_opencv as olk

    m, n = shape
    tex = _texture(max(m, 64), max(n, 64), seed=m + n)[:m, :n]
    frames = np.stack([tex, np.roll(tex, 1, axis=1)])
    got = dense_lk(frames)
    want = olk.dense_lucaskanade(frames)
    assert got.shape == (2, m, n) and np.isfinite(got).all()
    gxy, guv = dense_lk(frames, dense=False)
    wxy, wuv = olk.dense_lucaskanade(frames, dense=False)
    assert abs(len(gxy) - len(wxy)) <= max(2, 0.05 * len(wxy))
    if len(wxy) and len(gxy) == len(wxy):
        assert np.abs(got - want).max() < 1e-2


def test_flat_and_constant_frames(dense_lk):
    flat = np.full((2, 96, 96), 3.25, dtype=np.float32)
    assert np.all(dense_lk(flat) == 0)
    half = flat.copy()
    half[:, :, :40] = np.nan
    assert np.all(dense_lk(half) == 0)


def test_config2_2048_three_frames(dense_lk):
    """BASELINE config 2: 2048^2, 3 input frames (two frame pairs pooled), device resident: the
    estimate gives the known motion back and equals the estimate from host arrays."""
    from pysteps_amd import _lib, extrapolation
    from pysteps_amd.device import DeviceArray
    from tools import synth

    m = n = 2048
    base = synth.rain_field_db(m, n)
    vel = synth.true_velocity(m, n)
    vel_d = DeviceArray.from_host(vel)
    frames = DeviceArray((3, m, n), np.float32)
    lib = _lib.lib()
    _lib.check(lib.psh_memcpy_h2d(frames.ptr, base.ctypes.data, base.nbytes))
    adv = extrapolation.get_method("semilagrangian")(frames.view(0), vel_d, 2, outval=-15.0)
    _lib.check(lib.psh_memcpy_d2d(frames.view(1).ptr, adv.ptr, 2 * base.nbytes))
    field = dense_lk(frames).to_host()
    inner = (slice(None), slice(256, m - 256), slice(256, n - 256))
    rmse = np.sqrt(np.mean((field - vel)[inner] ** 2))
    assert rmse < 0.25, rmse
    host = dense_lk(frames.to_host())
    assert host.dtype == np.float64 and np.max(np.abs(host - field)) < 1e-5
    xy, uv = dense_lk(frames.to_host(), dense=False)
    assert 1000 < len(xy) <= 2000  # two pairs x <= 1000 corners


# the option matrix of pysteps/tests/test_motion_lk.py:20-40 (fd_method "shitomasi" rows)
LK_ARG_VALUES = [
    (True, 3, 30, 3, 20, False),   # defaults
    (False, 3, 30, 3, 20, True),   # sparse output, verbose
    (False, 0, 30, 3, 20, False),  # sparse output, all outliers
    (True, 3, None, 0, 0, False),  # global outlier detection, no filtering, no declustering
    (True, 0, 30, 3, 20, False),   # all outliers
]


@pytest.mark.parametrize("dense,nr_std_outlier,k_outlier,size_opening,decl_scale,verbose", LK_ARG_VALUES)
def test_lk_option_matrix(dense_lk, dense, nr_std_outlier, k_outlier, size_opening, decl_scale, verbose, capsys):
    """Formats as asserted by the reference's test_lk, and values against the oracle with the same
    options (three frames, two pooled pairs)."""
    from oracle import lk_opencv as olk

    frames, _ = _advected_frames(240, 272, 3, seed=23)
    out = dense_lk(frames, dense=dense, nr_std_outlier=nr_std_outlier, k_outlier=k_outlier,
                   size_opening=size_opening, decl_scale=decl_scale, verbose=verbose)
    want = olk.dense_lucaskanade(frames, dense=dense, nr_std_outlier=nr_std_outlier, k_outlier=k_outlier,
                                 size_opening=size_opening, decl_scale=decl_scale)
    if verbose:
        assert "Lucas-Kanade" in capsys.readouterr().out  # the reference prints its banner
    if dense:
        assert isinstance(out, np.ndarray) and out.ndim == 3 and out.shape == (2,) + frames.shape[1:]
        if nr_std_outlier == 0:
            assert out.sum() == 0
        else:
            # pixels whose k-th and (k+1)-th nearest vectors are equidistant pick either one (the
            # reference's cKDTree too): field-level agreement, pixel-level for all but those
            # (without declustering the same feature position appears once per frame pair: exact
            # distance ties between its copies, broken either way)
            loose = decl_scale == 0
            assert rel_l2(out, want) < (3e-3 if loose else 1e-3)
            assert np.mean(np.abs(out - want) > 1e-3) < (0.05 if loose else 0.01)
    else:
        assert isinstance(out, tuple) and len(out) == 2
        xy, uv = out
        assert xy.ndim == 2 and uv.ndim == 2 and xy.shape[1] == 2 and uv.shape == xy.shape
        if nr_std_outlier == 0:
            assert xy.shape[0] == 0
        else:
            assert abs(len(xy) - len(want[0])) <= max(3, 0.03 * len(want[0]))  # outlier test at 3 sigma
            wmap = {}
            for p, v in zip(want[0].astype(int), want[1]):
                wmap.setdefault(tuple(p), []).append(v)
            d = [min(np.abs(v - w).max() for w in wmap[tuple(p)]) for p, v in zip(xy.astype(int), uv)
                 if tuple(p) in wmap]
            assert len(d) >= 0.95 * len(want[0]) and max(d) < 1e-2


@pytest.mark.parametrize("fd_method,fd_kwargs", [("tstorm", None), ("blob", {"method": "doh"})])
def test_other_feature_detectors_are_delegated(dense_lk, fd_method, fd_kwargs):
    """The thunderstorm-cell detector and determinant-of-Hessian blobs are not part of the path (Shi-Tomasi and LoG / DoG
    blobs are: tests/test_blob_gpu.py): they go to the reference when it is importable and fail loudly otherwise (no
    silent substitution)."""
    frames, _ = _advected_frames(128, 128, 2, seed=5)
    try:  # with oracle/_ref importable the reference raises its own MissingOptionalDependency (skimage)
        from pysteps.exceptions import MissingOptionalDependency as missing
    except Exception:
        missing = NotImplementedError
    with pytest.raises((NotImplementedError, ImportError, ModuleNotFoundError, missing)):
        dense_lk(frames, fd_method=fd_method, fd_kwargs=fd_kwargs)


def test_other_interpolation_method_runs_on_the_hip_sparse_stage(dense_lk, ref_pysteps):
    """interp_method="rbfinterp2d" (SURVEY 8a row a12): features, tracking and outlier removal on the
    HIP path, declustering + the reference's own interpolation function (scipy.interpolate.Rbf behind
    pysteps/utils/interpolate.py:117-170) afterwards, as pysteps/motion/lucaskanade.py:264-274."""
    from pysteps.utils.cleansing import decluster as ref_decluster
    from pysteps.utils.interpolate import rbfinterp2d

    m = n = 192
    frames, vel = _advected_frames(m, n, 2, seed=8)
    got = dense_lk(frames, interp_method="rbfinterp2d", interp_kwargs={"epsilon": 5.0})
    assert got.shape == (2, m, n) and got.dtype == np.float64
    xy, uv = dense_lk(frames, dense=False)
    dxy, duv = ref_decluster(xy, uv, 20, 1)
    want = rbfinterp2d(dxy, duv, np.arange(n), np.arange(m), epsilon=5.0)
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9)
    inner = (slice(None), slice(m // 4, 3 * m // 4), slice(n // 4, 3 * n // 4))
    assert np.sqrt(np.mean((got - vel)[inner] ** 2)) < 1.0
    with pytest.raises(ValueError):
        dense_lk(frames, interp_method="no_such_interpolator")
    # nothing to interpolate -> zero field, whatever the method
    flat = np.zeros((2, 64, 64), dtype=np.float32)
    assert not dense_lk(flat, interp_method="rbfinterp2d").any()


@pytest.mark.parametrize("shape,nan", [((96, 130), False), ((257, 64), True), ((300, 300), True), ((1024, 1000), False)])
def test_prepare_float64_frames_bit_exact(lkmod, shape, nan):
    """float64 frames (what pysteps passes as a rule) are cleaned and quantised in double, like the
    reference does for them (utils/images.py:58-86, tracking/lucaskanade.py:135-160,
    shitomasi.py:128-151): both uint8 renderings bit-exact against the oracle run in float64 - and
    NOT what the float32 pipeline gives for the same frame rounded to float32 (the reason the
    float64 path exists: a grey level flips for ~2e-5 of the pixels)."""
    from oracle import lk_opencv as olk
    from pysteps_amd.device import DeviceArray

    m, n = shape
    rng = np.random.default_rng(n + 5)
    img = _rain(m, n, seed=m).astype(np.float64) + rng.uniform(-1e-7, 1e-7, (m, n))  # not float32 numbers
    speck = rng.random((m, n)) < 0.02
    img[speck] = rng.uniform(0, 30, speck.sum())
    if nan:
        img[: m // 5, : n // 3] = np.nan
        img[m // 2, n // 2] = np.nan
    valid = np.isfinite(img)
    prep = lkmod.PreparedFrame(DeviceArray.from_host(img), 3, 5, True)
    clean = olk.morph_opening(img, valid, img[valid].min())
    assert clean.dtype == np.float64
    got_clean = prep.clean.to_host()
    assert got_clean.dtype == np.float32 and np.array_equal(np.isnan(got_clean), ~valid)
    assert np.array_equal(got_clean[valid], clean[valid].astype(np.float32))
    lo, hi = clean[valid].min(), clean[valid].max()
    want_trk = olk.to_uint8(clean, valid, lo, hi, lo)
    assert np.array_equal(prep.track_u8.to_host(), want_trk)
    use = valid.copy()
    use[0, :] = False
    if (~valid).any():
        use[1, :] = False
    flo, fhi = clean[use].min(), clean[use].max()
    assert np.array_equal(prep.feature_u8.to_host(), olk.to_uint8(clean, use, flo, fhi, lo))
    if m * n >= 1 << 20:  # the float32 pipeline differs from this on a few pixels of a frame this size
        prep32 = _prep(lkmod, img.astype(np.float32))
        assert np.count_nonzero(prep32.track_u8.to_host() != want_trk) > 0


def test_dense_lk_float64_frames_match_the_oracle(dense_lk):
    """End to end with float64 input (the dtype of pysteps arrays): sparse vectors and dense field
    against the restatement run on the same float64 frames."""
    from oracle import lk_opencv as olk

    m = n = 320
    frames32, vel = _advected_frames(m, n, 2, seed=31)
    rng = np.random.default_rng(2)
    frames = frames32.astype(np.float64) + rng.uniform(-1e-6, 1e-6, frames32.shape)
    wxy, wuv = olk.dense_lucaskanade(frames, dense=False)
    gxy, guv = dense_lk(frames, dense=False)
    assert np.array_equal(gxy, wxy) and np.abs(guv - wuv).max() < 1e-2
    got, want = dense_lk(frames), olk.dense_lucaskanade(frames)
    assert got.dtype == np.float64 and rel_l2(got, want) < 1e-3
    # the staged Python loop takes the same path
    lk = __import__("pysteps_amd.motion.lucaskanade", fromlist=["x"])
    lk.USE_NATIVE_ORCHESTRATION = False
    try:
        sxy, suv = dense_lk(frames, dense=False)
    finally:
        lk.USE_NATIVE_ORCHESTRATION = True
    assert np.array_equal(sxy, gxy) and np.array_equal(suv, guv)


@pytest.mark.parametrize("max_corners,min_distance", [(1000, 7.5), (1000, 10.4), (300, 25), (3000, 6), (2500, 0)])
def test_corner_options_match_oracle(lkmod, max_corners, min_distance):
    """goodFeaturesToTrack options away from the defaults: a min_distance that is not an integer
    (OpenCV's cell size is round(min_distance), only the 3x3 cells around a candidate are searched),
    a large one, none at all, and more corners than the device walk keeps in LDS (> 2048: the
    candidates are ordered and walked on the host instead) - identical lists, identical order."""
    from oracle import lk_opencv as olk

    m, n = 640, 768
    img = _rain(m, n, seed=77)
    valid = np.isfinite(img)
    clean = olk.morph_opening(img, valid, img[valid].min())
    want = olk.shitomasi_detection(clean, valid, max_corners=max_corners, min_distance=min_distance)
    got = lkmod.detect_corners(_prep(lkmod, img), max_corners=max_corners, min_distance=min_distance)
    assert len(want) > 100
    assert np.array_equal(got, want)


def test_device_path_against_the_reference_orchestration_around_a_standin_cv2(dense_lk, ref_pysteps, tmp_path):
    """The device ``dense_lucaskanade`` against the REAL reference function (default options, three frames with a NaN
    block), the reference run in a process of its own around a stand-in ``cv2`` made of the restated OpenCV algorithms
    (tests/helpers/ref_lk_with_standin_cv2.py): vectors within 1e-2 px, dense field within 1e-3 relative L2 - the
    bars of the oracle comparison, here with the reference's own glue on the other side."""
    import os
    import subprocess
    import sys

    helper = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "ref_lk_with_standin_cv2.py")
    out = str(tmp_path / "ref_lk.npz")
    run = subprocess.run([sys.executable, helper, out], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    ref = np.load(out)
    frames, wxy, wuv, want = ref["frames"], ref["xy"], ref["uv"], ref["field"]
    gxy, guv = dense_lk(frames, dense=False)
    assert abs(len(gxy) - len(wxy)) <= max(3, 0.03 * len(wxy))
    wmap = {}
    for p, v in zip(wxy.astype(int), wuv):
        wmap.setdefault(tuple(p), []).append(v)
    d = [min(np.abs(v - w).max() for w in wmap[tuple(p)]) for p, v in zip(gxy.astype(int), guv) if tuple(p) in wmap]
    assert len(d) >= 0.95 * len(wxy) and max(d) < 1e-2
    got = dense_lk(frames)
    assert got.shape == want.shape and rel_l2(got, want) < 1e-3
