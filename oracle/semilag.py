"""CPU oracle for the semi-Lagrangian extrapolator.  TEST INFRASTRUCTURE ONLY.

This file is the checker for the HIP path, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  ``pysteps_amd`` must never import anything under ``oracle/``.

It restates, in float64 NumPy, the algorithm of the reference

    pysteps/extrapolation/semilagrangian.py:21-266   (``extrapolate``)
    pysteps/extrapolation/semilagrangian.py:181-198  (``interpolate_motion``)

whose arithmetic lives in the third-party ``scipy.ndimage.map_coordinates``
(SciPy 1.15.3 here; not under /root/reference).  Two samplers are provided:

* ``backend="numpy"``  - a self-contained restatement of order-0/1
  ``map_coordinates`` semantics: ``mode="nearest"``
  clamps the coordinate to ``[0, len-1]``; ``mode="constant"`` returns ``cval``
  iff a coordinate is ``< 0`` or ``> len-1`` (strict), otherwise the four taps
  ``floor, floor+1`` are all multiplied (a NaN tap poisons the sample even at
  weight 0) and a ``floor+1 == len`` tap is index-mirrored to ``len-2``.
* ``backend="scipy"``  - the same driver calling ``map_coordinates`` itself,
  i.e. the reference's own third-party kernel (used as the CPU baseline that
  is closest to the reference, and to validate the NumPy restatement).

Pinning: ``tests/test_oracle_semilag.py`` checks both backends against the two
known-answer tests of pysteps/tests/test_extrapolation_semilagrangian.py:9-24,
57-72 and against golden vectors produced by the real reference
(``tools/make_golden.py``, fixtures under ``tests/golden/``).
"""

import numpy as np

__all__ = ["extrapolate", "sample_velocity", "sample_field"]


# --------------------------------------------------------------------------
# order-0 / order-1 resampling, restated
# --------------------------------------------------------------------------
def _taps(coord, length):
    """floor index, mirrored upper index and upper weight for in-range coords."""
    base = np.floor(coord)
    frac = coord - base
    lo = base.astype(np.int64)
    hi = lo + 1
    # the upper tap can only leave the array when coord == length-1 exactly
    # (weight 0 there); SciPy mirrors that index back inside.
    hi = np.where(hi > length - 1, np.maximum(length - 2, 0), hi)
    lo = np.clip(lo, 0, length - 1)
    return lo, hi, frac


def _bilinear(field, row, col):
    m, n = field.shape
    r0, r1, fr = _taps(row, m)
    c0, c1, fc = _taps(col, n)
    f = field.astype(np.float64, copy=False)
    # all four products are formed, so NaN/Inf taps propagate at weight 0
    return (
        (1.0 - fr) * (1.0 - fc) * f[r0, c0]
        + (1.0 - fr) * fc * f[r0, c1]
        + fr * (1.0 - fc) * f[r1, c0]
        + fr * fc * f[r1, c1]
    )


# spline_filter's boundary initialisation per map_coordinates mode (ni_splines.c `apply_filter`), pinned
# against scipy.ndimage.spline_filter1d itself (tests/test_oracle_semilag.py): "constant", "mirror",
# "wrap" and "grid-constant" filter with whole-sample MIRROR boundaries, "nearest" and "reflect" with
# half-sample REFLECT boundaries, "grid-wrap" periodically
_PREFILTER_KIND = {"constant": "mirror", "mirror": "mirror", "wrap": "mirror", "grid-constant": "mirror",
                   "nearest": "reflect", "reflect": "reflect", "grid-wrap": "wrap"}
# "nearest" and "grid-constant" have no exact boundary condition in the filter: map_coordinates pads the
# array by 12 samples (edge values / cval) first and shifts the coordinates (_prepad_for_spline_filter)
_PREFILTER_PAD = {"nearest": 12, "grid-constant": 12}


def _spline_poles(order):
    """Poles of the B-spline prefilter (ni_splines.c get_filter_poles; order 3: the one of the cubic case)."""
    if order == 2:
        return [np.sqrt(8.0) - 3.0]
    if order == 3:
        return [np.sqrt(3.0) - 2.0]
    if order == 4:
        return [np.sqrt(664.0 - np.sqrt(438976.0)) + np.sqrt(304.0) - 19.0,
                np.sqrt(664.0 + np.sqrt(438976.0)) - np.sqrt(304.0) - 19.0]
    if order == 5:
        return [np.sqrt(67.5 - np.sqrt(4436.25)) + np.sqrt(26.25) - 6.5,
                np.sqrt(67.5 + np.sqrt(4436.25)) - np.sqrt(26.25) - 6.5]
    raise NotImplementedError("spline order %r" % (order,))


def _spline_prefilter(field, kind="mirror", order=3):
    """B-spline coefficients of order 2 .. 5 along both axes, float64 (scipy.ndimage.spline_filter; ni_splines.c
    apply_filter: the gain prod (1 - z)(1 - 1/z) over the poles on the whole line first - 6 for the cubic case -, then
    per pole _init_causal_* / causal recursion / _init_anticausal_* / anticausal recursion of the boundary kind)."""
    poles = _spline_poles(order)
    gain = 1.0
    for z in poles:
        gain *= (1.0 - z) * (1.0 - 1.0 / z)
    c = np.array(field, dtype=np.float64)
    for axis in (0, 1):
        c = np.moveaxis(c, axis, 0).copy()
        n = c.shape[0]
        if n > 1:
            c *= gain
            for z in poles:
                _spline_pole_pass(c, n, z, kind)
        c = np.moveaxis(c, 0, axis)
    return c


def _spline_pole_pass(c, n, z, kind):
    if kind == "mirror":
        zn1 = z ** (n - 1)
        acc = c[0] + zn1 * c[n - 1]
        zi = z
        for i in range(1, n - 1):
            acc = acc + zi * (c[i] + zn1 * c[n - 1 - i])
            zi *= z
        c[0] = acc / (1.0 - zn1 * zn1)
    elif kind == "reflect":
        zn = z**n
        first = c[0].copy()
        acc = c[0] + zn * c[n - 1]
        zi = z
        for i in range(1, n):
            acc = acc + zi * (c[i] + zn * c[n - 1 - i])
            zi *= z
        c[0] = acc * (z / (1.0 - zn * zn)) + first
    else:  # wrap
        acc = c[0].copy()
        zi = z
        for i in range(1, n):
            acc = acc + zi * c[n - i]
            zi *= z
        c[0] = acc / (1.0 - zi)
    for i in range(1, n):
        c[i] += z * c[i - 1]
    if kind == "mirror":
        c[n - 1] = (z * c[n - 2] + c[n - 1]) * z / (z * z - 1.0)
    elif kind == "reflect":
        c[n - 1] = c[n - 1] * (z / (z - 1.0))
    else:
        acc = c[n - 1].copy()
        zi = z
        for i in range(0, n - 1):
            acc = acc + zi * c[i]
            zi *= z
        c[n - 1] = acc * (z / (zi - 1.0))
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])


def _spline_prefilter_mirror(field):
    return _spline_prefilter(field, "mirror")


def _mirror_index(i, n):
    if n == 1:
        return np.zeros_like(i)
    period = 2 * (n - 1)
    i = np.mod(i, period)
    return np.where(i >= n, period - i, i)


def _bspline3_weights(t):
    return [(1 - t) ** 3 / 6, (3 * t**3 - 6 * t**2 + 4) / 6, (-3 * t**3 + 3 * t**2 + 3 * t + 1) / 6, t**3 / 6]


def _bspline_basis(x, order):
    """Centred cardinal B-spline of the given order at distance x (the closed piecewise polynomials)."""
    a = np.abs(x)
    if order == 2:
        return np.where(a <= 0.5, 0.75 - a * a, np.where(a <= 1.5, 0.5 * (1.5 - a) ** 2, 0.0))
    if order == 3:
        return np.where(a <= 1.0, 2.0 / 3.0 - a * a + 0.5 * a**3, np.where(a <= 2.0, (2.0 - a) ** 3 / 6.0, 0.0))
    if order == 4:
        return np.where(a <= 0.5, 115.0 / 192.0 - 0.625 * a * a + 0.25 * a**4,
                        np.where(a <= 1.5, 55.0 / 96.0 + a * (5.0 / 24.0 + a * (-1.25 + a * (5.0 / 6.0 - a / 6.0))),
                                 np.where(a <= 2.5, (2.5 - a) ** 4 / 24.0, 0.0)))
    if order == 5:
        return np.where(a <= 1.0, 0.55 + a * a * (-0.5 + a * a * (0.25 - a / 12.0)),
                        np.where(a <= 2.0, 0.425 + a * (0.625 + a * (-1.75 + a * (1.25 + a * (-0.375 + a / 24.0)))),
                                 np.where(a <= 3.0, (3.0 - a) ** 5 / 120.0, 0.0)))
    raise NotImplementedError("spline order %r" % (order,))


def _spline_taps(c, order):
    """(first tap index, list of the order + 1 weights) of map_coordinates for coordinates c: odd orders start at
    floor(c) - order // 2, even ones at floor(c + 0.5) - order // 2 (ni_interpolation.c); the weights are the
    B-spline at the taps' distances (SciPy sets the last one to 1 - sum of the others: equal to rounding)."""
    base = np.floor(c) if order & 1 else np.floor(c + 0.5)
    start = base.astype(np.int64) - order // 2
    return start, [_bspline_basis(c - (start + k), order) for k in range(order + 1)]


def _cubic(field, row, col, cval, order=3):
    """order-2 .. 5 map_coordinates, mode="constant": strict inside test on the coordinate, (order + 1)^2 taps with
    mirrored indices, coefficients from the mirror prefilter."""
    m, n = field.shape
    coef = _spline_prefilter(field, "mirror", order)
    outside = (row < 0.0) | (row > m - 1.0) | (col < 0.0) | (col > n - 1.0)
    rr, cc = np.where(outside, 0.0, row), np.where(outside, 0.0, col)
    if order == 3:
        iy, ix = np.floor(rr).astype(np.int64) - 1, np.floor(cc).astype(np.int64) - 1
        wy, wx = _bspline3_weights(rr - iy - 1), _bspline3_weights(cc - ix - 1)
    else:
        iy, wy = _spline_taps(rr, order)
        ix, wx = _spline_taps(cc, order)
    acc = np.zeros(row.shape)
    for a in range(order + 1):
        for b in range(order + 1):
            acc += wy[a] * wx[b] * coef[_mirror_index(iy + a, m), _mirror_index(ix + b, n)]
    return np.where(outside, cval, acc)


# ---- boundary modes other than "constant" (map_coordinates_mode, reference :91-96, :225-232) ----
# Restated from the behaviour of SciPy 1.15.3 and pinned against it on dense probes with NaNs
# planted at every index (tests/test_oracle_semilag.py::test_boundary_modes_*):
#  1. the coordinate is folded into the array for "mirror", "reflect", "wrap", "grid-wrap"
#     ("nearest" and "grid-constant" leave it alone; "constant" only decides inside/outside);
#  2. taps are floor(c), floor(c)+1 (order 1) or floor(c+0.5) (order 0);
#  3. a tap outside [0,len) is clamped ("nearest"), reflected ("reflect"), taken modulo len
#     ("grid-wrap"), replaced by cval ("grid-constant"), and MIRRORED for "mirror", "wrap", "constant".
def _trunc_div(a, b):
    """C-style (npy_intp)(a / b) for float a, int b."""
    return np.trunc(a / b)


def _fold_coordinate(x, length, mode):
    x = np.array(x, dtype=np.float64, copy=True)
    if mode in ("nearest", "grid-constant", "constant"):
        return x
    if length <= 1:
        return np.where((x < 0) | (x > length - 1), 0.0, x)
    lo, hi = x < 0, x > length - 1
    out = x.copy()
    if mode == "mirror":
        s2 = 2 * length - 2
        t = s2 * _trunc_div(-x, s2) + x
        out = np.where(lo, np.where(t <= 1 - length, t + s2, -t), out)
        t = x - s2 * _trunc_div(x, s2)
        out = np.where(hi, np.where(t >= length, s2 - t, t), out)
    elif mode == "reflect":
        s2 = 2 * length
        t = np.where(x < -s2, s2 * _trunc_div(-x, s2) + x, x)
        out = np.where(lo, np.where(t < -length, t + s2, -t - 1), out)
        t = x - s2 * _trunc_div(x, s2)
        out = np.where(hi, np.where(t >= length, s2 - t - 1, t), out)
    elif mode == "wrap":
        s1 = length - 1
        out = np.where(lo, x + s1 * (_trunc_div(-x, s1) + 1), out)
        out = np.where(hi, x - s1 * _trunc_div(x, s1), out)
    elif mode == "grid-wrap":
        out = np.where(lo, x + length * (_trunc_div(-1 - x, length) + 1), out)
        out = np.where(hi, x - length * _trunc_div(x + 1, length), out)
    else:
        raise RuntimeError("boundary mode not supported")
    return out


def _fold_tap(i, length, mode):
    """(index, is_cval) of integer taps ``i`` that may lie outside [0, length)."""
    i = np.array(i, dtype=np.int64, copy=True)
    outside = (i < 0) | (i >= length)
    if mode == "grid-constant":
        return np.where(outside, 0, i), outside
    no = np.zeros(i.shape, dtype=bool)
    if length <= 1:
        return np.zeros_like(i), no
    if mode == "nearest":
        return np.clip(i, 0, length - 1), no
    if mode == "grid-wrap":
        return np.mod(i, length), no
    if mode == "reflect":
        s2 = 2 * length
        t = np.where(i < -s2, i + s2 * ((-i) // s2), i)
        neg = np.where(t < -length, t + s2, -t - 1)
        t = i - s2 * (np.maximum(i, 0) // s2)
        pos = np.where(t >= length, s2 - t - 1, t)
    else:  # mirror, wrap, constant
        s2 = 2 * length - 2
        t = i + s2 * ((-i) // s2)
        neg = np.where(t <= 1 - length, t + s2, -t)
        t = i - s2 * (np.maximum(i, 0) // s2)
        pos = np.where(t >= length, s2 - t, t)
    return np.where(i < 0, neg, np.where(i >= length, pos, i)), no


def _cubic_mode(field, row, col, mode, cval, order=3):
    """order-2 .. 5 map_coordinates with a boundary mode other than "constant": the array is padded for the
    two modes the filter has no boundary condition for, filtered with the mode's boundary kind, the
    coordinate folded like for the lower orders (on the ORIGINAL length) and shifted by the padding, the
    (order + 1)^2 taps folded index by index on the padded length ("grid-constant": cval)."""
    m, n = field.shape
    npad = _PREFILTER_PAD.get(mode, 0)
    if mode == "nearest":
        padded = np.pad(field, npad, mode="edge")
    elif mode == "grid-constant":
        padded = np.pad(field, npad, mode="constant", constant_values=cval)
    else:
        padded = field
    coef = _spline_prefilter(padded, _PREFILTER_KIND[mode], order)
    rr = _fold_coordinate(row, m, mode) + npad
    cc = _fold_coordinate(col, n, mode) + npad
    big_m, big_n = padded.shape
    if order == 3:
        iy, ix = np.floor(rr).astype(np.int64) - 1, np.floor(cc).astype(np.int64) - 1
        wy, wx = _bspline3_weights(rr - iy - 1), _bspline3_weights(cc - ix - 1)
    else:
        iy, wy = _spline_taps(rr, order)
        ix, wx = _spline_taps(cc, order)
    acc = np.zeros(np.shape(row))
    for a in range(order + 1):
        ri, rcv = _fold_tap(iy + a, big_m, mode)
        for b in range(order + 1):
            ci, ccv = _fold_tap(ix + b, big_n, mode)
            acc = acc + wy[a] * wx[b] * np.where(rcv | ccv, cval, coef[ri, ci])
    return acc


def _numpy_sample_mode(field, row, col, mode, cval, order):
    m, n = field.shape
    f = field.astype(np.float64, copy=False)
    rr = _fold_coordinate(row, m, mode)
    cc = _fold_coordinate(col, n, mode)
    if order == 0:
        ri, rc = _fold_tap(np.floor(rr + 0.5).astype(np.int64), m, mode)
        ci, cv = _fold_tap(np.floor(cc + 0.5).astype(np.int64), n, mode)
        return np.where(rc | cv, cval, f[ri, ci])
    r0 = np.floor(rr)
    c0 = np.floor(cc)
    fr, fc = rr - r0, cc - c0
    acc = 0.0
    for dr, wr in ((0, 1.0 - fr), (1, fr)):
        ri, rcv = _fold_tap(r0.astype(np.int64) + dr, m, mode)
        for dc, wc in ((0, 1.0 - fc), (1, fc)):
            ci, ccv = _fold_tap(c0.astype(np.int64) + dc, n, mode)
            acc = acc + wr * wc * np.where(rcv | ccv, cval, f[ri, ci])
    return acc


def _numpy_sample(field, row, col, mode, cval, order):
    m, n = field.shape
    row = np.asarray(row, dtype=np.float64)
    col = np.asarray(col, dtype=np.float64)
    # Non-finite coordinates (trajectories that met a non-finite velocity, allow_nonfinite_values):
    # SciPy 1.15 answers with cval in mode "constant" (any order) and with NaN where it
    # interpolates across them in mode "nearest" (order >= 1) - pinned by the sl_velnan* goldens of
    # the unmodified reference.  The other modes are not restated for such coordinates.
    bad = ~(np.isfinite(row) & np.isfinite(col))
    if bad.any():
        if mode == "constant":
            lost = cval
        elif mode == "nearest" and order >= 1:
            lost = np.nan
        else:
            raise NotImplementedError("non-finite coordinates are restated for modes constant / nearest only")
        val = _numpy_sample(field, np.where(bad, 0.0, row), np.where(bad, 0.0, col), mode, cval, order)
        return np.where(bad, lost, val)
    if mode != "constant" and order in (2, 3, 4, 5):
        return _cubic_mode(field.astype(np.float64), row, col, mode, cval, order)
    if mode != "constant" and order in (0, 1) and not (mode == "nearest" and np.all(np.isfinite(field))):
        return _numpy_sample_mode(field, row, col, mode, cval, order)
    if mode == "nearest":
        # finite data: clamping the coordinate gives the same values as clamping the taps
        rr = np.clip(row, 0.0, m - 1.0)
        cc = np.clip(col, 0.0, n - 1.0)
        outside = None
    elif mode == "constant":
        outside = (row < 0.0) | (row > m - 1.0) | (col < 0.0) | (col > n - 1.0)
        rr = np.where(outside, 0.0, row)
        cc = np.where(outside, 0.0, col)
    else:
        raise NotImplementedError("numpy backend restates order 3 for mode constant only")

    if order in (2, 3, 4, 5):
        if mode != "constant":
            raise NotImplementedError("numpy backend restates the spline orders for mode constant here")
        return _cubic(field.astype(np.float64), row, col, cval, order)
    if order == 1:
        val = _bilinear(field, rr, cc)
    elif order == 0:
        ri = np.floor(rr + 0.5).astype(np.int64)
        ci = np.floor(cc + 0.5).astype(np.int64)
        val = field.astype(np.float64, copy=False)[
            np.clip(ri, 0, m - 1), np.clip(ci, 0, n - 1)
        ]
    else:
        raise NotImplementedError("numpy backend restates interpolation orders 0 .. 5")

    if outside is not None:
        val = np.where(outside, cval, val)
    return val


def _scipy_sample(field, row, col, mode, cval, order):
    from scipy.ndimage import map_coordinates

    return map_coordinates(
        field, [row, col], mode=mode, cval=cval, order=order, prefilter=order > 1
    )


_BACKENDS = {"numpy": _numpy_sample, "scipy": _scipy_sample}


def sample_velocity(velocity, dx, dy, backend="numpy"):
    """Both velocity components at (x+dx, y+dy), edge-clamped (reference :181-190)."""
    m, n = velocity.shape[1:]
    yy, xx = np.mgrid[0:m, 0:n]
    fn = _BACKENDS[backend]
    return np.stack(
        [fn(velocity[c], yy + dy, xx + dx, "nearest", 0.0, 1) for c in range(2)]
    )


def sample_field(field, dx, dy, cval=np.nan, order=1, backend="numpy", mode="constant"):
    """Scalar field at (x+dx, y+dy); outside -> cval (reference :221-232)."""
    m, n = field.shape
    yy, xx = np.mgrid[0:m, 0:n]
    return _BACKENDS[backend](field, yy + dy, xx + dx, mode, cval, order)


# --------------------------------------------------------------------------
# driver
# --------------------------------------------------------------------------
def _step_sizes(timesteps, vel_timestep):
    """Lead-time increments in units of the velocity time step (reference :159-165)."""
    if isinstance(timesteps, (int, np.integer)) and not isinstance(timesteps, bool):
        return np.ones(int(timesteps), dtype=np.float64)
    ts = np.asarray(timesteps, dtype=np.float64)
    if ts.ndim != 1 or ts.size == 0:
        raise ValueError("timesteps must be an int or a 1-d sequence")
    if np.any(np.diff(ts) <= 0.0):
        raise ValueError("the given timestep sequence is not monotonously increasing")
    return np.concatenate([ts[:1], np.diff(ts)]) / float(vel_timestep)


def extrapolate(
    precip,
    velocity,
    timesteps,
    outval=np.nan,
    xy_coords=None,
    allow_nonfinite_values=False,
    vel_timestep=1,
    displacement_prev=None,
    n_iter=1,
    return_displacement=False,
    interp_order=1,
    backend="numpy",
    map_coordinates_mode="constant",
    verbose=False,
    D_prev=None,
):
    """float64 restatement of semilagrangian.extrapolate (reference :21-266).

    Positional order as the reference's (``outval`` 4th, ``xy_coords`` 5th - steps.py:697-703
    passes "min" positionally and ``xy_coords=`` by name); ``xy_coords`` replaces the default
    integer meshgrid (:174-179); ``verbose`` / deprecated ``D_prev`` are accepted and ignored.

    Returns what the reference returns: ``(T,m,n)`` in the dtype of ``precip``
    (SciPy allocates its output in the input dtype), optionally with the
    float64 displacement ``(2,m,n)``; ``(None, displacement)`` for precip None.
    """
    sampler = _BACKENDS[backend]
    if precip is not None and precip.ndim != 2:
        raise ValueError("precip must be a two-dimensional array")
    if velocity.ndim != 3:
        raise ValueError("velocity must be a three-dimensional array")
    if not allow_nonfinite_values:
        if precip is not None and not np.all(np.isfinite(precip)):
            raise ValueError("precip contains non-finite values")
        if not np.all(np.isfinite(velocity)):
            raise ValueError("velocity contains non-finite values")
    if precip is not None and not np.any(np.isfinite(precip)):
        raise ValueError("precip contains only non-finite values")
    if not np.any(np.isfinite(velocity)):
        raise ValueError("velocity contains only non-finite values")
    if isinstance(timesteps, list) and sorted(timesteps) != timesteps:
        raise ValueError("timesteps is not in ascending order")
    if precip is None and not return_displacement:
        raise ValueError("precip is None but return_displacement is False")

    steps = _step_sizes(timesteps, vel_timestep)
    if isinstance(outval, str):
        if outval != "min":
            raise ValueError("outval must be a number or 'min'")
        outval = np.nanmin(precip) if precip is not None else np.nan

    m, n = velocity.shape[1:]
    if xy_coords is None:
        yy, xx = np.mgrid[0:m, 0:n]
    else:
        xx, yy = np.asarray(xy_coords)[0], np.asarray(xy_coords)[1]  # [0] = x/cols, [1] = y/rows (:256-258)
    sub = float(n_iter) if n_iter > 1 else 1.0

    # interp_order > 1: the field is interpolated with NaNs zeroed and two order-1 mask warps
    # restore the no-rain minimum and the missing values (reference :146-157, :234-253)
    if precip is not None and interp_order > 1:
        minval = np.nanmin(precip)
        mask_min = (precip > minval).astype(float)
        if allow_nonfinite_values:
            mask_finite = np.isfinite(precip)
            precip = precip.copy()
            precip[~mask_finite] = 0.0
            mask_finite = mask_finite.astype(float)
        else:
            mask_finite = np.ones(precip.shape)

    def motion_at(dx, dy, step):
        # map_coordinates allocates its output in the dtype of the sampled
        # array, so float32 velocities give float32-rounded samples.
        v = np.stack(
            [
                np.asarray(
                    sampler(velocity[c], yy + dy, xx + dx, "nearest", 0.0, 1)
                ).astype(velocity.dtype)
                for c in range(2)
            ]
        ).astype(np.float64)
        return v / sub * step

    if displacement_prev is None:
        disp = np.zeros((2, m, n))
        # NB the very first increment is NOT divided by n_iter (reference :202)
        inc = velocity.astype(np.float64) * steps[0]
        resumed = False
    else:
        disp = np.array(displacement_prev, dtype=np.float64)
        inc = motion_at(disp[0], disp[1], steps[0])
        resumed = True

    frames = []
    for ti, step in enumerate(steps):
        if n_iter > 0:
            for _ in range(n_iter):
                inc = motion_at(disp[0] - inc[0] / 2.0, disp[1] - inc[1] / 2.0, step)
                disp = disp - inc
                inc = motion_at(disp[0], disp[1], step)
        else:
            if ti > 0 or resumed:
                inc = motion_at(disp[0], disp[1], step)
            disp = disp - inc

        if precip is not None:
            val = np.asarray(
                sampler(precip, yy + disp[1], xx + disp[0], map_coordinates_mode, outval, interp_order)
            )
            val = val.astype(precip.dtype, copy=True)
            if interp_order > 1:
                # both masks with the caller's boundary mode (reference :234-253)
                warped = sampler(mask_min, yy + disp[1], xx + disp[0], map_coordinates_mode, 0, 1)
                val[np.asarray(warped) < 0.5] = minval
                warped = sampler(mask_finite, yy + disp[1], xx + disp[0], map_coordinates_mode, 0, 1)
                val[np.asarray(warped) < 0.5] = np.nan
            frames.append(val)

    if precip is None:
        return None, disp
    out = np.stack(frames)
    return (out, disp) if return_displacement else out
