"""CPU oracle for the OpenCV front end of dense Lucas-Kanade.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED AT THE OPENCV BOUNDARY.  The arithmetic of this part of the
reference lives in a third-party dependency that is absent from /root/reference
and not installed here: OpenCV (``opencv-python``, version unpinned in the
reference's requirements.txt:2 / environment.yml).  The five call sites are

    pysteps/utils/images.py:72,75          getStructuringElement(MORPH_ELLIPSE,(3,3)), morphologyEx(OPEN)
    pysteps/feature/shitomasi.py:137       dilate(mask, ones(5,5))
    pysteps/feature/shitomasi.py:162       goodFeaturesToTrack(maxCorners=1000, q=0.01, minDist=10, block=5)
    pysteps/tracking/lucaskanade.py:171    calcOpticalFlowPyrLK(winSize=(50,50), maxLevel=3, criteria=(3,10,0),
                                                               minEigThreshold=1e-4)

This module restates the published OpenCV 4.x algorithms behind those calls
(modules/imgproc/src/{morph,corner,featureselect,pyramids}.cpp,
modules/video/src/lkpyramid.cpp; notes in SURVEY.md section 8a) in NumPy, plus the
NumPy glue of the reference around them (images.py:58-86, shitomasi.py:122-171,
tracking/lucaskanade.py:130-189, motion/lucaskanade.py:182-279).  No golden
vectors exist for it (the reference's LK tests need cv2 and skip here); it is
anchored by the reference's *property* tests instead (tests/test_lk_oracle.py:
uniform-shift recovery < 0.1 % rel. RMSE like pysteps/tests/test_motion.py:154-250,
zeros -> zero motion, NaN vs masked equivalence) and re-validation against real
cv2 is required whenever a box with OpenCV is available.

What IS pinned (tests/test_lk_oracle.py, all on the CPU):
* the NumPy glue and the orchestration - the REAL reference ``dense_lucaskanade`` run around a stand-in ``cv2``
  whose five functions are the ones below gives this module's pipeline bit for bit (sparse vectors equal, dense
  fields identical; tests/helpers/ref_lk_with_standin_cv2.py);
* every restated algorithm against a second, independent reading built from SciPy: opening and mask dilation
  (binary_erosion / binary_dilation), pyrDown and Scharr (correlate1d, mirror), cornerMinEigenVal (sobel +
  uniform_filter), the selection stage of goodFeaturesToTrack (maximum_filter + all-pairs distances, identical
  lists), the pyramidal tracker (float64 Lucas-Kanade with map_coordinates: 1e-4 px).
Two authors and the reference's own glue - not OpenCV itself: the header stays "unpinned at the OpenCV boundary".

Deliberate choices where OpenCV's own result is build dependent (SIMD summation
order): window sums (A11, A12, A22, b1, b2) are accumulated exactly in integers
and converted once; the 5x5 box sums of the corner response are accumulated in
float64 (boxFilter's sum type for 32F data) and rounded to float32.
"""

import numpy as np

# --------------------------------------------------------------------------
# small image helpers
# --------------------------------------------------------------------------
def _reflect101(idx, n):
    """BORDER_REFLECT_101 index map (gfedcb|abcdefgh|gfedcba)."""
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.mod(idx, period)
    return np.where(idx >= n, period - idx, idx)


def _pad_reflect101(img, r):
    m, n = img.shape
    ys = _reflect101(np.arange(-r, m + r), m)
    xs = _reflect101(np.arange(-r, n + r), n)
    return img[np.ix_(ys, xs)]


def masked_min_max(img):
    ok = np.isfinite(img)
    vals = img[ok]
    return vals.min(), vals.max()


def to_uint8(img, valid, lo, hi, fill):
    """(img - lo) / (hi - lo) * 255 truncated to uint8, invalid pixels = fill
    (tracking/lucaskanade.py:143-160, shitomasi.py:143-151); float32 arithmetic
    for float32 frames, like NumPy does in the reference."""
    dt = img.dtype if img.dtype in (np.float32, np.float64) else np.float64
    filled = np.where(valid, img, fill).astype(dt)
    lo, hi = dt.type(lo), dt.type(hi)
    if (hi - lo) > 1e-8:
        scaled = (filled - lo) / (hi - lo) * 255
    else:
        scaled = filled - lo
    return scaled.astype(np.uint8)


# --------------------------------------------------------------------------
# utils/images.py:27-86  morph_opening  (cv2.morphologyEx OPEN, 3x3 cross)
# --------------------------------------------------------------------------
def _structuring_element_ellipse(n):
    """cv2.getStructuringElement(MORPH_ELLIPSE, (n, n)) (morph.cpp): row i spans
    |x - c| <= round(c * sqrt(1 - dy^2/r^2)) around the centre; (3,3) is the plus."""
    r, c = n // 2, n // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    k = np.zeros((n, n), dtype=bool)
    for i in range(n):
        dy = i - r
        if abs(dy) <= r:
            dx = int(round(c * np.sqrt(max((r * r - dy * dy) * inv_r2, 0.0))))
            j1, j2 = max(c - dx, 0), min(c + dx + 1, n)
            k[i, j1:j2] = True
    return k


def _morph(binary, kernel, erode):
    """Binary erode/dilate with the border treated as neutral (OpenCV default)."""
    m, n = binary.shape
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    pad = np.full((m + kh - 1, n + kw - 1), erode, dtype=bool)
    pad[ay:ay + m, ax:ax + n] = binary
    out = np.full((m, n), erode, dtype=bool)
    for i in range(kh):
        for j in range(kw):
            if kernel[i, j]:
                win = pad[i:i + m, j:j + n]
                out = (out & win) if erode else (out | win)
    return out


def morph_opening(img, valid, thr, n=3):
    """Pixels > thr that do not survive a binary opening are set to the minimum."""
    fill = img[valid].min()
    filled = np.where(valid, img, fill)
    field = filled > thr
    kernel = _structuring_element_ellipse(n)
    opened = _morph(_morph(field, kernel, True), kernel, False)
    removed = field & ~opened
    out = img.copy()
    out[removed] = fill
    return out


# --------------------------------------------------------------------------
# feature/shitomasi.py:26-171  (cv2.dilate, cv2.goodFeaturesToTrack)
# --------------------------------------------------------------------------
def dilate_mask(mask, size):
    """cv2.dilate(mask, ones((size,size))): anchor at the kernel centre size//2."""
    if size <= 0:
        return mask.copy()
    m, n = mask.shape
    a = size // 2
    pad = np.zeros((m + size - 1, n + size - 1), dtype=bool)
    pad[a:a + m, a:a + n] = mask
    out = np.zeros((m, n), dtype=bool)
    for i in range(size):
        for j in range(size):
            out |= pad[i:i + m, j:j + n]
    return out


def corner_min_eigenval(u8, block_size=5):
    """cv::cornerMinEigenVal(src 8U, blockSize, ksize=3), float32 result."""
    s = np.float32(1.0 / (4.0 * block_size * 255.0))
    p = _pad_reflect101(u8.astype(np.float32), 1)
    m, n = u8.shape
    # Sobel dx: [-1 0 1] along x, [1 2 1]*s along y;  dy: transposed roles
    hx = p[:, 2:] - p[:, :-2]                        # (m+2, n) exact integers
    dx = (hx[:-2] + hx[2:]) * s + hx[1:-1] * (np.float32(2) * s)
    hy = (p[:, :-2] + p[:, 2:]) * s + p[:, 1:-1] * (np.float32(2) * s)
    dy = hy[2:] - hy[:-2]
    dx, dy = dx.astype(np.float32), dy.astype(np.float32)
    r = block_size // 2

    def box(a):
        q = _pad_reflect101(a, r).astype(np.float64)
        acc = np.zeros((m, n), dtype=np.float64)
        for i in range(block_size):
            for j in range(block_size):
                acc += q[i:i + m, j:j + n]
        return acc.astype(np.float32)

    cxx, cxy, cyy = box(dx * dx), box(dx * dy), box(dy * dy)
    a, b, c = cxx * np.float32(0.5), cxy, cyy * np.float32(0.5)
    return ((a + c) - np.sqrt((a - c) * (a - c) + b * b)).astype(np.float32)


def good_features_to_track(u8, allowed, max_corners=1000, quality=0.01, min_distance=10, block_size=5):
    """cv::goodFeaturesToTrack (featureselect.cpp), Shi-Tomasi variant.  Returns (p,2) float32 (x,y)."""
    m, n = u8.shape
    eig = corner_min_eigenval(u8, block_size)
    if not allowed.any():
        return np.empty((0, 2), dtype=np.float32)
    max_val = eig[allowed].max()
    thr = np.float32(max_val * quality)
    eig = np.where(eig > thr, eig, np.float32(0))            # THRESH_TOZERO
    pad = np.full((m + 2, n + 2), -np.inf, dtype=np.float32)  # dilate: border neutral
    pad[1:-1, 1:-1] = eig
    dil = np.max([pad[i:i + m, j:j + n] for i in range(3) for j in range(3)], axis=0)
    cand = (eig != 0) & (eig == dil) & allowed
    cand[0, :] = cand[-1, :] = False
    cand[:, 0] = cand[:, -1] = False
    ys, xs = np.nonzero(cand)
    vals = eig[ys, xs]
    # descending value, ties by descending address (greaterThanPtr)
    order = np.lexsort((-(ys * n + xs), -vals.astype(np.float64)))
    ys, xs = ys[order], xs[order]
    if min_distance < 1:
        sel = slice(None, max_corners if max_corners > 0 else None)
        return np.column_stack([xs[sel], ys[sel]]).astype(np.float32)
    cell = int(round(min_distance))
    gw, gh = (n + cell - 1) // cell, (m + cell - 1) // cell
    grid = [[[] for _ in range(gw)] for _ in range(gh)]
    md2 = float(min_distance) ** 2
    out = []
    for y, x in zip(ys.tolist(), xs.tolist()):
        xc, yc = x // cell, y // cell
        good = True
        for yy in range(max(0, yc - 1), min(gh - 1, yc + 1) + 1):
            for xx in range(max(0, xc - 1), min(gw - 1, xc + 1) + 1):
                for (px, py) in grid[yy][xx]:
                    if (x - px) ** 2 + (y - py) ** 2 < md2:
                        good = False
                        break
                if not good:
                    break
            if not good:
                break
        if good:
            grid[yc][xc].append((x, y))
            out.append((x, y))
            if 0 < max_corners == len(out):
                break
    return np.array(out, dtype=np.float32).reshape(-1, 2)


def shitomasi_detection(img, valid, max_corners=1000, quality_level=0.01, min_distance=10,
                        block_size=5, buffer_mask=5):
    """feature/shitomasi.py:122-171 on an (m,n) frame with validity mask."""
    fill = img[valid].min()                       # fill value fixed before the row masking (:131)
    mask = ~valid
    use = valid.copy()
    if buffer_mask > 0:
        mask = dilate_mask(mask, int(buffer_mask))
        # shitomasi.py:140 indexes with the uint8 mask: row 0 always, row 1 if any pixel is masked
        use[0, :] = False
        if mask.any() and img.shape[0] > 1:
            use[1, :] = False
    lo, hi = img[use].min(), img[use].max()
    u8 = to_uint8(img, use, lo, hi, fill)
    return good_features_to_track(u8, ~mask, max_corners, quality_level, min_distance, block_size)


# --------------------------------------------------------------------------
# tracking/lucaskanade.py:35-189  (cv2.calcOpticalFlowPyrLK)
# --------------------------------------------------------------------------
def pyr_down(u8):
    """cv::pyrDown for 8U: separable [1 4 6 4 1], reflect-101, (sum + 128) >> 8."""
    m, n = u8.shape
    om, on = (m + 1) // 2, (n + 1) // 2
    src = u8.astype(np.int32)
    xs = 2 * np.arange(on)
    w = (1, 4, 6, 4, 1)
    rows = sum(wk * src[:, _reflect101(xs + k - 2, n)] for k, wk in enumerate(w))
    ys = 2 * np.arange(om)
    acc = sum(wk * rows[_reflect101(ys + k - 2, m), :] for k, wk in enumerate(w))
    return ((acc + 128) >> 8).astype(np.uint8)


def scharr_deriv(u8):
    """calcSharrDeriv (lkpyramid.cpp): int16 (Ix, Iy), reflect-101, gain 32."""
    p = _pad_reflect101(u8.astype(np.int32), 1)
    t0 = (p[:-2] + p[2:]) * 3 + p[1:-1] * 10      # vertical smoothing, (m, n+2)
    t1 = p[2:] - p[:-2]                           # vertical difference
    ix = t0[:, 2:] - t0[:, :-2]
    iy = (t1[:, 2:] + t1[:, :-2]) * 3 + t1[:, 1:-1] * 10
    return ix.astype(np.int16), iy.astype(np.int16)


def build_pyramid(u8, win, max_level):
    """buildOpticalFlowPyramid: stop before a level is not larger than the window."""
    levels = [u8]
    for _ in range(max_level):
        nxt = pyr_down(levels[-1])
        if nxt.shape[1] <= win[0] or nxt.shape[0] <= win[1]:
            break
        levels.append(nxt)
    return levels


def _descale(v, n):
    return (v + (1 << (n - 1))) >> n


def _cv_round(v):
    return np.rint(v).astype(np.int64)  # cvRound: round half to even (SSE cvtss2si)


def _weights(a, b):
    a, b = np.float32(a), np.float32(b)
    one = np.float32(1)
    s = np.float32(1 << 14)
    iw00 = int(_cv_round((one - a) * (one - b) * s))
    iw01 = int(_cv_round(a * (one - b) * s))
    iw10 = int(_cv_round((one - a) * b * s))
    return iw00, iw01, iw10, (1 << 14) - iw00 - iw01 - iw10


def _window(padded, border, ix, iy, w, h):
    """(h+1, w+1) block whose top-left is image pixel (ix, iy) in a `border`-padded array."""
    return padded[iy + border:iy + border + h + 1, ix + border:ix + border + w + 1]


def _bilinear_int(block, iw, shift):
    iw00, iw01, iw10, iw11 = iw
    v = block[:-1, :-1] * iw00 + block[:-1, 1:] * iw01 + block[1:, :-1] * iw10 + block[1:, 1:] * iw11
    return _descale(v, shift)


def calc_optical_flow_pyr_lk(prev_u8, next_u8, points, win=(50, 50), max_level=3, max_count=10,
                             epsilon=0.0, min_eig_threshold=1e-4):
    """cv::calcOpticalFlowPyrLK restated (LKTrackerInvoker).  points (p,2) float32 (x,y).
    Returns next_points (p,2) float32 and status (p,) bool."""
    f32 = np.float32
    w, h = int(win[0]), int(win[1])
    pts = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    npts = pts.shape[0]
    prev_pyr = build_pyramid(prev_u8, (w, h), max_level)
    next_pyr = build_pyramid(next_u8, (w, h), max_level)
    top = min(len(prev_pyr), len(next_pyr)) - 1
    nxt = np.zeros_like(pts)
    status = np.ones(npts, dtype=bool)
    half = (f32((w - 1) * 0.5), f32((h - 1) * 0.5))
    eps2 = f32(epsilon) * f32(epsilon)
    flt_scale = f32(1.0 / (1 << 20))
    flt_eps = f32(np.finfo(np.float32).eps)
    border = max(w, h) + 2
    for level in range(top, -1, -1):
        I, J = prev_pyr[level], next_pyr[level]
        rows, cols = I.shape
        Ipad = _pad_reflect101(I.astype(np.int64), border)
        Jpad = _pad_reflect101(J.astype(np.int64), border)
        dx, dy = scharr_deriv(I)
        dxp = np.zeros((rows + 2 * border, cols + 2 * border), dtype=np.int64)
        dyp = np.zeros_like(dxp)
        dxp[border:-border, border:-border] = dx
        dyp[border:-border, border:-border] = dy
        scale = f32(1.0 / (1 << level))
        for i in range(npts):
            prev_pt = (pts[i, 0] * scale, pts[i, 1] * scale)
            if level == top:
                next_pt = prev_pt
            else:
                next_pt = (nxt[i, 0] * f32(2), nxt[i, 1] * f32(2))
            nxt[i] = next_pt
            ppx, ppy = f32(prev_pt[0] - half[0]), f32(prev_pt[1] - half[1])
            ipx, ipy = int(np.floor(ppx)), int(np.floor(ppy))
            if ipx < -w or ipx >= cols or ipy < -h or ipy >= rows:
                if level == 0:
                    status[i] = False
                continue
            iw = _weights(ppx - f32(ipx), ppy - f32(ipy))
            Ipatch = _bilinear_int(_window(Ipad, border, ipx, ipy, w, h), iw, 14 - 5)
            gx = _bilinear_int(_window(dxp, border, ipx, ipy, w, h), iw, 14)
            gy = _bilinear_int(_window(dyp, border, ipx, ipy, w, h), iw, 14)
            A11 = f32(int((gx * gx).sum())) * flt_scale
            A12 = f32(int((gx * gy).sum())) * flt_scale
            A22 = f32(int((gy * gy).sum())) * flt_scale
            D = f32(A11 * A22) - f32(A12 * A12)
            disc = f32(f32(f32(A11 - A22) * f32(A11 - A22)) + f32(f32(4) * A12) * A12)
            min_eig = f32(f32(A22 + A11) - np.sqrt(disc, dtype=np.float32)) / f32(2 * w * h)
            if min_eig < min_eig_threshold or D < flt_eps:
                if level == 0:
                    status[i] = False
                continue
            D = f32(1) / D
            npx, npy = f32(next_pt[0] - half[0]), f32(next_pt[1] - half[1])
            prev_delta = (f32(0), f32(0))
            for j in range(max_count):
                inx, iny = int(np.floor(npx)), int(np.floor(npy))
                if inx < -w or inx >= cols or iny < -h or iny >= rows:
                    if level == 0:
                        status[i] = False
                    break
                iwj = _weights(npx - f32(inx), npy - f32(iny))
                diff = _bilinear_int(_window(Jpad, border, inx, iny, w, h), iwj, 14 - 5) - Ipatch
                b1 = f32(int((diff * gx).sum())) * flt_scale
                b2 = f32(int((diff * gy).sum())) * flt_scale
                ddx = f32(f32(f32(A12 * b2) - f32(A22 * b1)) * D)
                ddy = f32(f32(f32(A12 * b1) - f32(A11 * b2)) * D)
                npx, npy = f32(npx + ddx), f32(npy + ddy)
                nxt[i] = (f32(npx + half[0]), f32(npy + half[1]))
                if f32(ddx * ddx) + f32(ddy * ddy) <= eps2:
                    break
                if j > 0 and abs(ddx + prev_delta[0]) < 0.01 and abs(ddy + prev_delta[1]) < 0.01:
                    nxt[i] = (f32(nxt[i, 0] - ddx * f32(0.5)), f32(nxt[i, 1] - ddy * f32(0.5)))
                    break
                prev_delta = (ddx, ddy)
            if level == 0 and status[i]:
                # err is always requested by the Python binding: final window must start inside
                fx, fy = f32(nxt[i, 0] - half[0]), f32(nxt[i, 1] - half[1])
                rx, ry = int(_cv_round(fx)), int(_cv_round(fy))
                if rx < -w or rx >= cols or ry < -h or ry >= rows:
                    status[i] = False
    return nxt, status


def track_features(prev_img, next_img, prev_valid, next_valid, points, winsize=(50, 50), nr_levels=3,
                   max_count=10, epsilon=0.0, min_eig_thr=1e-4):
    """tracking/lucaskanade.py:130-189: per-frame min-max scaling to uint8, then pyramidal LK."""
    lo, hi = prev_img[prev_valid].min(), prev_img[prev_valid].max()
    p8 = to_uint8(prev_img, prev_valid, lo, hi, lo)
    lo, hi = next_img[next_valid].min(), next_img[next_valid].max()
    n8 = to_uint8(next_img, next_valid, lo, hi, lo)
    p1, st = calc_optical_flow_pyr_lk(p8, n8, points, winsize, nr_levels, max_count, epsilon, min_eig_thr)
    if st.any():
        return points[st], p1[st] - points[st]
    return np.empty((0, 2)), np.empty((0, 2))


# --------------------------------------------------------------------------
# motion/lucaskanade.py:182-279  sparse vectors of dense_lucaskanade
# --------------------------------------------------------------------------
def sparse_lucaskanade(frames, size_opening=3, buffer_mask=5, max_corners=1000, quality_level=0.01,
                       min_distance=10, block_size=5, winsize=(50, 50), nr_levels=3, min_eig_thr=1e-4, detector=None):
    """Pooled (xy, uv) before outlier removal (lucaskanade.py:205-242).  `detector`: another feature detection
    method (lucaskanade.py:191,230: called with the frame after the opening, missing pixels NaN) -> (p, 2) points."""
    xy = np.empty((0, 2))
    uv = np.empty((0, 2))
    for t in range(frames.shape[0] - 1):
        prev, nxt = frames[t].copy(), frames[t + 1].copy()
        pv, nv = np.isfinite(prev), np.isfinite(nxt)
        if size_opening > 0:
            prev = morph_opening(prev, pv, prev[pv].min(), size_opening)
            nxt = morph_opening(nxt, nv, nxt[nv].min(), size_opening)
        if detector is not None:
            pts = np.asarray(detector(prev)).astype(np.float32)
        else:
            pts = shitomasi_detection(prev, pv, max_corners, quality_level, min_distance, block_size, buffer_mask)
        if pts.shape[0] == 0:
            continue
        xy_, uv_ = track_features(prev, nxt, pv, nv, pts, winsize, nr_levels, 10, 0.0, min_eig_thr)
        if xy_.shape[0] == 0:
            continue
        xy = np.append(xy, xy_, axis=0)
        uv = np.append(uv, uv_, axis=0)
    return xy, uv


def dense_lucaskanade(frames, dense=True, nr_std_outlier=3, k_outlier=30, decl_scale=20, **kw):
    """Full restated pipeline (lucaskanade.py:182-279) -> (2,m,n) float64 or (xy, uv)."""
    from . import sparse as osp

    m, n = frames.shape[1:]
    xy, uv = sparse_lucaskanade(frames, **kw)
    if xy.shape[0] == 0:
        return np.zeros((2, m, n)) if dense else (xy, uv)
    out = osp.detect_outliers(uv, nr_std_outlier, xy, k_outlier)
    xy, uv = xy[~out], uv[~out]
    if not dense:
        return xy, uv
    if decl_scale > 1:
        xy, uv = osp.decluster(xy, uv, decl_scale, 1)
    if xy.shape[0] == 0:
        return np.zeros((2, m, n))
    return osp.idw(xy, uv, m, n)
