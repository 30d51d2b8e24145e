"""The device-resident sparse stage of dense LK (csrc/lk_sparse.hip) against the host entry points.

corner_order (psh_lk_order_host) has to reproduce, for ANY candidate set, what sorting the keys
descending and walking them with psh_lk_greedy_host gives (that host pass is pinned against the
OpenCV restatement in tests/test_lk_oracle.py) - including the cases the histogram selection has
to work for: more candidates than one chunk, walks that need several chunks, thousands of equal
responses (the threshold is then found inside a bin, through the address bits), min_distance that
is not an integer (OpenCV's cell rounding), min_distance < 1, max_corners cutting a batch.
vectors_finish (psh_vectors_finish_host) against psh_decluster_host and the interpolator preamble
of pysteps/decorators.py:199-208.  Index work: everything here is bit-exact.
"""

import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)


def _keys(resp, addr):
    return (_bits(resp) << np.uint64(32)) | addr.astype(np.uint64)


def _host_walk(lib, keys, m, n, min_distance, max_corners):
    order = np.sort(keys)[::-1].copy()
    pts = np.empty((max_corners, 2), np.float32)
    count = ctypes.c_int(0)
    rc = lib.psh_lk_greedy_host(order.ctypes.data, int(order.size), m, n, float(min_distance), int(max_corners),
                                pts.ctypes.data, ctypes.byref(count))
    assert rc == 0
    return pts[: count.value].copy()


def _device_walk(lib, keys, top, quality, m, n, min_distance, max_corners):
    from pysteps_amd import _lib

    pts = np.empty((max_corners, 2), np.float32)
    count = ctypes.c_int(0)
    _lib.check(lib.psh_lk_order_host(keys.ctypes.data, int(keys.size), float(top), float(quality), m, n,
                                     float(min_distance), int(max_corners), pts.ctypes.data, ctypes.byref(count)),
               "psh_lk_order_host")
    return pts[: count.value].copy()


CASES = [
    # name, N, response law, spatial law, (m, n), min_distance, max_corners
    ("small", 300, "cubic", "uniform", (512, 640), 10, 1000),
    ("one_chunk", 4096, "cubic", "uniform", (2048, 2048), 10, 1000),
    ("many_candidates", 150000, "cubic", "uniform", (4096, 4096), 10, 1000),
    ("huge", 1200000, "cubic", "uniform", (4096, 4096), 10, 1000),
    ("several_chunks", 60000, "cubic", "clustered", (4096, 4096), 10, 1000),  # few corners fit: the walk goes on
    ("walks_everything", 30000, "cubic", "clustered", (2048, 2048), 25, 2000),
    ("all_equal", 50000, "equal", "uniform", (4096, 4096), 10, 1000),  # threshold inside one bin (address bits)
    ("three_values", 50000, "three", "uniform", (4096, 4096), 10, 500),
    ("fractional_down", 40000, "cubic", "uniform", (3000, 3000), 10.4, 1000),  # cell = round(10.4) = 10 < distance
    ("fractional_up", 40000, "cubic", "uniform", (3000, 3000), 9.6, 1000),
    ("no_distance", 20000, "cubic", "uniform", (1024, 1024), 0.0, 777),
    ("distance_one", 20000, "cubic", "clustered", (1024, 1024), 1.0, 1500),
    ("cut_in_batch", 9000, "cubic", "uniform", (4096, 4096), 3, 61),
    ("tiny_image", 900, "cubic", "uniform", (30, 30), 4, 2048),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_corner_order_matches_sorted_host_walk(case):
    from pysteps_amd import _lib

    name, N, law, spatial, (m, n), min_distance, max_corners = case
    lib = _lib.lib()
    rng = np.random.default_rng(len(name) * 1000 + N)
    top = np.float32(3.7)
    quality = 0.01
    if law == "equal":
        resp = np.full(N, 1.25, np.float32)
    elif law == "three":
        resp = rng.choice(np.array([0.5, 1.0, 3.7], np.float32), N)
    else:
        resp = (top * (0.0101 + 0.9899 * rng.random(N) ** 3)).astype(np.float32)
    resp[0] = top
    if spatial == "clustered":  # candidates crowd a small square: most are rejected
        side = int(min(m, n, max(60, np.sqrt(N) * 1.2)))
        sel = rng.permutation(side * side)[:N]
        y, x = np.divmod(sel, side)
        addr = (y + (m - side) // 2) * n + (x + (n - side) // 2)
    else:
        addr = rng.permutation(m * n)[:N]
    assert len(np.unique(addr)) == N
    keys = _keys(resp, addr)
    thr = np.float32(top * np.float32(quality))
    assert (resp > thr).all() and (resp <= top).all()
    want = _host_walk(lib, keys, m, n, min_distance, max_corners)
    got = _device_walk(lib, keys, top, quality, m, n, min_distance, max_corners)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want)


def test_corner_order_no_candidates_and_limits():
    from pysteps_amd import _lib

    lib = _lib.lib()
    keys = np.empty(0, np.uint64)
    got = _device_walk(lib, keys, 1.0, 0.01, 64, 64, 10, 100)
    assert got.shape == (0, 2)
    pts = np.empty((4096, 2), np.float32)
    count = ctypes.c_int(0)
    rc = lib.psh_lk_order_host(keys.ctypes.data, 0, 1.0, 0.01, 64, 64, 10.0, 4096, pts.ctypes.data, ctypes.byref(count))
    assert rc == _lib.PSH_EUNSUPPORTED  # more corners than the kernel's LDS list: the host pass takes those


def _host_finish(lib, xy, uv, flags, scale, m, n):
    """psh_decluster_host + the preamble as dense_lucaskanade / the interpolator decorator apply it."""
    keep = np.ones(len(xy), bool) if len(xy) < 2 else ~flags.astype(bool)
    xy, uv = np.ascontiguousarray(xy[keep]), np.ascontiguousarray(uv[keep])
    if len(xy) and scale > 1:
        oxy, ouv = np.empty_like(xy), np.empty_like(uv)
        count = ctypes.c_int(0)
        assert lib.psh_decluster_host(xy.ctypes.data, uv.ctypes.data, len(xy), float(scale), 1, oxy.ctypes.data,
                                      ouv.ctypes.data, ctypes.byref(count)) == 0
        xy, uv = oxy[: count.value], ouv[: count.value]
    if len(xy) == 0:
        return xy, uv, 1, (0.0, 0.0)
    if len(xy) == 1:
        return xy, uv, 1, (np.float32(uv[0, 0]), np.float32(uv[0, 1]))
    if uv.max() == uv.min():
        return xy, uv, 1, (np.float32(uv[0, 0]), np.float32(uv[0, 0]))
    return xy, uv, 0, None


def _device_finish(lib, xy, uv, flags, scale, m, n):
    from pysteps_amd import _lib

    cap = max(len(xy), 1)
    oxy, ouv = np.empty((cap, 2), np.float32), np.empty((cap, 2), np.float32)
    count, mode = ctypes.c_int(0), ctypes.c_int(-1)
    const = np.zeros(2, np.float32)
    reach = np.zeros(1, np.float32)
    _lib.check(lib.psh_vectors_finish_host(xy.ctypes.data, uv.ctypes.data, flags.ctypes.data, len(xy), float(scale),
                                           m, n, oxy.ctypes.data, ouv.ctypes.data, ctypes.byref(count),
                                           ctypes.byref(mode), const.ctypes.data, reach.ctypes.data),
               "psh_vectors_finish_host")
    return oxy[: count.value], ouv[: count.value], mode.value, const, float(reach[0])


@pytest.mark.parametrize("nvec,scale,flag_rate,spread", [
    (1000, 20.0, 0.03, 4096), (3000, 20.0, 0.1, 2048), (8192, 20.0, 0.0, 4096), (2500, 7.5, 0.05, 700),
    (900, 1.0, 0.08, 4096), (900, 0.0, 0.5, 512), (4000, 300.0, 0.02, 4096), (700, 1e6, 0.0, 4096),
    (2, 20.0, 0.5, 100), (1, 20.0, 1.0, 100), (0, 20.0, 0.0, 100), (5, 20.0, 1.0, 100),
])
def test_vectors_finish_matches_host_decluster(nvec, scale, flag_rate, spread):
    from pysteps_amd import _lib

    lib = _lib.lib()
    m = n = 4096
    rng = np.random.default_rng(nvec + int(scale * 10))
    # corner positions are integer valued, vectors are float32 differences (tracking/lucaskanade.py:181)
    xy = rng.integers(0, spread, size=(nvec, 2)).astype(np.float64)
    uv = (rng.standard_normal((nvec, 2)) * 3).astype(np.float32).astype(np.float64)
    if nvec > 10:
        uv[rng.integers(0, nvec, nvec // 5)] = uv[0]  # equal values: the medians have to break ties alike
    flags = (rng.random(nvec) < flag_rate).astype(np.uint8)
    wxy, wuv, wmode, wconst = _host_finish(lib, xy, uv, flags, scale, m, n)
    gxy, guv, gmode, gconst, greach = _device_finish(lib, xy, uv, flags, scale, m, n)
    assert gmode == wmode
    assert len(gxy) == len(wxy)
    if wmode == 1:
        assert np.array_equal(gconst, np.asarray(wconst, np.float32))
        return
    assert np.array_equal(gxy, wxy.astype(np.float32)) and np.array_equal(guv, wuv.astype(np.float32))
    dx = max(wxy[:, 0].max(), n - 1.0) - min(wxy[:, 0].min(), 0.0)
    dy = max(wxy[:, 1].max(), m - 1.0) - min(wxy[:, 1].min(), 0.0)
    assert greach == pytest.approx(np.hypot(dx, dy) * 1.001 + 1.0, rel=1e-6)


def test_vectors_finish_all_equal_and_single_cell():
    from pysteps_amd import _lib

    lib = _lib.lib()
    xy = np.array([[10, 10], [500, 40], [900, 900]], np.float64)
    uv = np.full((3, 2), 2.5)
    flags = np.zeros(3, np.uint8)
    gxy, guv, gmode, gconst, _ = _device_finish(lib, xy, uv, flags, 20.0, 1024, 1024)
    assert gmode == 1 and np.array_equal(gconst, np.float32([2.5, 2.5])) and len(gxy) == 3
    # every vector in one cell: the component-wise medians (even count: mean of the middle pair)
    xy = np.array([[3, 4], [5, 1], [8, 8], [2, 9]], np.float64)
    uv = np.array([[1, -1], [2, 7], [4, 0], [3, 5]], np.float64)
    gxy, guv, gmode, gconst, _ = _device_finish(lib, xy, uv, np.zeros(4, np.uint8), 20.0, 64, 64)
    assert gmode == 1 and len(gxy) == 1
    assert np.array_equal(gxy[0], np.float32([4.0, 6.0])) and np.array_equal(gconst, np.float32([2.5, 2.5]))
    assert np.array_equal(guv[0], np.float32([2.5, 2.5]))


def test_dense_estimate_is_asynchronous_and_repeatable():
    """Resident frames in, resident field out: the call only queues kernels (no count is asked
    for), so estimates issued back to back must not disturb each other's device-resident state."""
    from pysteps_amd.device import DeviceArray
    from pysteps_amd.motion import get_method
    from tools import synth

    m = n = 768
    base = synth.rain_field_db(m, n, seed=11, sigma=8.0)
    other = synth.rain_field_db(m, n, seed=12, sigma=8.0)
    fa = DeviceArray.from_host(np.stack([base, np.roll(base, (2, 3), axis=(0, 1))]), dtype=np.float32)
    fb = DeviceArray.from_host(np.stack([other, np.roll(other, (-1, 2), axis=(0, 1))]), dtype=np.float32)
    lk = get_method("LK")
    first_a = lk(fa).to_host()
    first_b = lk(fb).to_host()
    out = [lk(f) for f in (fa, fb, fa, fb, fa)]  # five estimates queued before anything is read
    assert np.array_equal(out[0].to_host(), first_a) and np.array_equal(out[2].to_host(), first_a)
    assert np.array_equal(out[4].to_host(), first_a)
    assert np.array_equal(out[1].to_host(), first_b) and np.array_equal(out[3].to_host(), first_b)
    host = lk(np.stack([base, np.roll(base, (2, 3), axis=(0, 1))]))
    assert host.dtype == np.float64 and np.array_equal(host.astype(np.float32), first_a)
