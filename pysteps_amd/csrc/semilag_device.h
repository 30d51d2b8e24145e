// Device helpers shared by the semi-Lagrangian kernels (semilag.hip, semilag_members.hip):
// split integer+fraction trajectory arithmetic and the order-0/1 resampling rules of
// scipy.ndimage.map_coordinates as used by pysteps/extrapolation/semilagrangian.py:181-232.
// Include AFTER `#pragma clang fp contract(off)`: floor(t) and t - floor(t) must see the
// same rounded t (see semilag.hip).
#pragma once

#include "common.h"

namespace psh {
namespace sl {

constexpr float kMaxFrac = 0x1.fffffep-1f;  // largest float below 1

// uniform base + 32-bit lane byte offset (+ small immediate):
//   global_load_dword v, v_off, s[base:base+1] offset:imm
__device__ __forceinline__ float ld(const float *base, unsigned byte_off, int elem = 0) {
  return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off)[elem];
}

// Position along one axis is an integer pixel index P plus a fraction f in [0,1).
// Subtract w: |rounding| <= ulp(|f - w|)/2 ~ 5e-7 px for |w| < 8, independent of
// how far the trajectory has travelled.
// Integer positions are moved by SATURATING adds (v_add_i32 / v_sub_i32 with the clamp bit, the cost of a plain add):
// a step the conversion saturated - a sentinel velocity such as 1e20 - parks the trajectory at INT_MAX / INT_MIN, far
// outside every image on the side it left on, instead of wrapping to the other side or, after a few such steps, back
// inside; the reference's float64 position behaves the same way.
__device__ __forceinline__ int sat_add(int a, int b) {
  int r;
  asm("v_add_i32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ int sat_sub(int a, int b) {
  int r;
  asm("v_sub_i32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ void retreat(int &P, float &f, float w) {
  const float t = f - w;
  int k;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k) : "v"(t));  // (int)floor(t), one instruction
  P = sat_add(P, k);
  // v_fract_f32 = min(t - floor(t), 0x1.fffffep-1): (-1e-9) - (-1) would round to 1.0f, the
  // instruction keeps f < 1 (checked on the device by tests/test_semilag_gpu.py::test_fraction_clamp)
  f = __builtin_amdgcn_fractf(t);
}

// float64 displacement of the reference -> split coordinate (P += floor(d), f = d - floor(d)).
// A non-finite displacement - a trajectory that met a non-finite velocity value
// (allow_nonfinite_values, semilagrangian.py:106-137) - gives a NaN fraction and leaves P alone:
// v_cvt_flr(NaN) = 0 keeps the integer part inside the image, every later sample is NaN, and the
// field sample is replaced by what map_coordinates returns for a NaN coordinate (`lost`).
__device__ __forceinline__ void split_displacement(double d, int &P, float &f) {
  const bool ok = fabs(d) < 1e300;  // false for NaN and +-inf
  const double fl = ok ? floor(d) : 0.0;
  P = sat_add(P, static_cast<int>(fmin(fmax(fl, -2147483648.0), 2147483647.0)));
  f = ok ? fminf(static_cast<float>(d - fl), kMaxFrac) : __builtin_nanf("");
}
__device__ __forceinline__ bool lost(float fx, float fy) { return fx != fx || fy != fy; }

// true when every active lane of the wave has all four taps strictly inside the image
__device__ __forceinline__ bool wave_all_interior(int X, int Y, int m, int n) {
  const unsigned long long bx = __builtin_amdgcn_ballot_w64(static_cast<unsigned>(X) < static_cast<unsigned>(n - 1));
  const unsigned long long by = __builtin_amdgcn_ballot_w64(static_cast<unsigned>(Y) < static_cast<unsigned>(m - 1));
  return (bx & by) == __builtin_amdgcn_ballot_w64(true);
}

struct Weights {
  float w00, w01, w10, w11;
};

__device__ __forceinline__ Weights make_weights(float fx, float fy) {
  const float gx = 1.f - fx, gy = 1.f - fy;
  return {gy * gx, gy * fx, fy * gx, fy * fx};
}

// all four products are formed: a NaN tap poisons the sample even at weight 0,
// exactly like map_coordinates
__device__ __forceinline__ float blend(const Weights &w, float a, float b, float c, float d) {
  return fmaf(w.w11, d, fmaf(w.w10, c, fmaf(w.w01, b, w.w00 * a)));
}

__device__ __forceinline__ bool is_interior(int X, int Y, int m, int n) {
  return static_cast<unsigned>(X) < static_cast<unsigned>(n - 1) &&
         static_cast<unsigned>(Y) < static_cast<unsigned>(m - 1);
}

// precip, mode="constant": outside (coord < 0 or > len-1, strict) -> outval; the
// upper tap at floor+1 == len is index-mirrored (weight 0 there).  Loads are
// unconditional (indices clamped) so that they can be issued with the velocity taps.
template <int ORDER>
__device__ __forceinline__ float sample_precip_border(const float *p, int X, int Y, float fx,
                                                      float fy, int m, int n, float outval) {
  const bool outside = X < 0 || Y < 0 || X > n - 1 || Y > m - 1 || (X == n - 1 && fx > 0.f) ||
                       (Y == m - 1 && fy > 0.f);
  const int xc = min(max(X, 0), n - 1), yc = min(max(Y, 0), m - 1);
  float val;
  if (ORDER == 0) {
    // floor(c + 0.5): half rounds up
    const int xi = min(xc + (fx >= 0.5f ? 1 : 0), n - 1);
    const int yi = min(yc + (fy >= 0.5f ? 1 : 0), m - 1);
    val = ld(p, static_cast<unsigned>(__mul24(yi, n) + xi) << 2);
  } else {
    const int x1 = (xc + 1 > n - 1) ? max(n - 2, 0) : xc + 1;
    const int y1 = (yc + 1 > m - 1) ? max(m - 2, 0) : yc + 1;
    const unsigned r0 = static_cast<unsigned>(__mul24(yc, n)), r1 = static_cast<unsigned>(__mul24(y1, n));
    const Weights w = make_weights(fx, fy);
    val = blend(w, ld(p, (r0 + xc) << 2), ld(p, (r0 + x1) << 2), ld(p, (r1 + xc) << 2),
                ld(p, (r1 + x1) << 2));
  }
  return outside ? outval : val;
}


// ---- map_coordinates_mode other than "constant" (interp_order 0 / 1) ------------------------------
// scipy.ndimage.map_coordinates (SciPy 1.15, the reference's resampler at
// pysteps/extrapolation/semilagrangian.py:225-232) restated on the split coordinate c = P + f,
// f in [0,1): first the coordinate is folded into the array by the mode ("mirror", "reflect",
// "wrap", "grid-wrap"; "nearest" and "grid-constant" leave it alone), then the taps floor(c),
// floor(c)+1 (order 1) or floor(c+0.5) (order 0) that fall outside [0,len) are folded index by
// index: clamped ("nearest"), reflected ("reflect"), taken modulo len ("grid-wrap"), replaced by
// cval ("grid-constant") and MIRRORED for "mirror", "wrap" and "constant".  The rules were pinned
// against SciPy itself on dense probes with NaNs planted at every index (oracle/semilag.py,
// tests/test_oracle_semilag.py); the integer conditions below are the exact images of SciPy's
// double comparisons because f < 1.
enum : int {
  kModeConstant = 0, kModeNearest = 1, kModeReflect = 2, kModeMirror = 3, kModeWrap = 4,
  kModeGridConstant = 5, kModeGridWrap = 6
};

// c -> -c
__device__ __forceinline__ void negate_coord(int &P, float &f) {
  if (f > 0.f) {
    P = -P - 1;
    f = fminf(1.f - f, kMaxFrac);
  } else {
    P = -P;
  }
}

// (npy_intp)(-c / s) for c < 0: truncation of a positive quotient
__device__ __forceinline__ int trunc_neg_over(int P, float f, int s) { return (f > 0.f ? -P - 1 : -P) / s; }

__device__ __forceinline__ void fold_coord(int &P, float &f, int len, int mode) {
  const bool below = P < 0, above = P > len - 1 || (P == len - 1 && f > 0.f);
  if (!(below || above) || mode == kModeNearest || mode == kModeGridConstant || mode == kModeConstant) return;
  if (len <= 1) {
    P = 0;
    f = 0.f;
    return;
  }
  if (mode == kModeMirror) {
    const int s2 = 2 * len - 2;
    if (below) {
      P += s2 * trunc_neg_over(P, f, s2);
      if (P < 1 - len || (P == 1 - len && f == 0.f)) P += s2; else negate_coord(P, f);
    } else {
      P -= s2 * (P / s2);
      if (P >= len) { negate_coord(P, f); P += s2; }
    }
  } else if (mode == kModeReflect) {
    const int s2 = 2 * len;
    if (below) {
      if (P < -s2) P += s2 * trunc_neg_over(P, f, s2);
      if (P < -len) P += s2; else { negate_coord(P, f); P -= 1; }
    } else {
      P -= s2 * (P / s2);
      if (P >= len) { negate_coord(P, f); P += s2 - 1; }
    }
  } else if (mode == kModeWrap) {
    const int s = len - 1;
    if (below) P += s * (trunc_neg_over(P, f, s) + 1); else P -= s * (P / s);
  } else {  // grid-wrap
    if (below) P += len * ((f > 0.f ? -P - 2 : -P - 1) / len + 1); else P -= len * ((P + 1) / len);
  }
}

// index of a tap that fell outside [0,len); *is_cval is set for "grid-constant"
__device__ __forceinline__ int fold_tap(int i, int len, int mode, bool *is_cval) {
  if (static_cast<unsigned>(i) < static_cast<unsigned>(len)) return i;
  if (mode == kModeGridConstant) {
    *is_cval = true;
    return 0;
  }
  if (len <= 1) return 0;
  if (mode == kModeNearest) return min(max(i, 0), len - 1);
  if (mode == kModeGridWrap) {
    const int r = i % len;
    return r < 0 ? r + len : r;
  }
  if (mode == kModeReflect) {
    const int s2 = 2 * len;
    if (i < 0) {
      if (i < -s2) i += s2 * (-i / s2);
      return i < -len ? i + s2 : -i - 1;
    }
    i -= s2 * (i / s2);
    return i >= len ? s2 - i - 1 : i;
  }
  const int s2 = 2 * len - 2;  // mirror, wrap, constant
  if (i < 0) {
    i += s2 * (-i / s2);
    return i <= 1 - len ? i + s2 : -i;
  }
  i -= s2 * (i / s2);
  return i >= len ? s2 - i : i;
}

template <int ORDER>
__device__ __forceinline__ float sample_precip_mode(const float *p, int X, int Y, float fx, float fy, int m,
                                                    int n, float cval, int mode) {
  fold_coord(X, fx, n, mode);
  fold_coord(Y, fy, m, mode);
  if (ORDER == 0) {
    bool cv = false;
    const int xi = fold_tap(X + (fx >= 0.5f ? 1 : 0), n, mode, &cv);
    const int yi = fold_tap(Y + (fy >= 0.5f ? 1 : 0), m, mode, &cv);
    const float v = ld(p, static_cast<unsigned>(__mul24(yi, n) + xi) << 2);
    return cv ? cval : v;
  }
  bool cx0 = false, cx1 = false, cy0 = false, cy1 = false;
  const int x0 = fold_tap(X, n, mode, &cx0), x1 = fold_tap(X + 1, n, mode, &cx1);
  const int y0 = fold_tap(Y, m, mode, &cy0), y1 = fold_tap(Y + 1, m, mode, &cy1);
  const unsigned r0 = static_cast<unsigned>(__mul24(y0, n)), r1 = static_cast<unsigned>(__mul24(y1, n));
  float a = ld(p, (r0 + x0) << 2), b = ld(p, (r0 + x1) << 2);
  float c = ld(p, (r1 + x0) << 2), d = ld(p, (r1 + x1) << 2);
  a = (cx0 || cy0) ? cval : a;
  b = (cx1 || cy0) ? cval : b;
  c = (cx0 || cy1) ? cval : c;
  d = (cx1 || cy1) ? cval : d;
  return blend(make_weights(fx, fy), a, b, c, d);
}

// the advected field off the clamp-free interior path, any boundary mode
template <int ORDER>
__device__ __forceinline__ float sample_precip_edge(const float *p, int X, int Y, float fx, float fy, int m,
                                                    int n, float outval, int mode) {
  return mode == kModeConstant ? sample_precip_border<ORDER>(p, X, Y, fx, fy, m, n, outval)
                               : sample_precip_mode<ORDER>(p, X, Y, fx, fy, m, n, outval, mode);
}

// ---- interp_order = 3 -------------------------------------------------------------------
// map_coordinates(order=3, mode="constant") on the prefiltered coefficients (csrc/spline.hip)
// plus the two order-1 mask warps of pysteps/extrapolation/semilagrangian.py:234-253: pixels
// whose warped "finite" mask is < 0.5 become NaN (this includes everything advected from
// outside, whatever outval is), pixels whose warped "above the minimum" mask is < 0.5 become
// the minimum.  Both masks are functions of the original field, so they are evaluated from its
// four order-1 taps instead of from two extra planes.
__device__ __forceinline__ int mirror101(int i, int n) {
  if (n == 1) return 0;
  if (i < 0) i = -i;
  if (i >= n) {
    const int period = 2 * (n - 1);
    i %= period;
    if (i >= n) i = period - i;
  }
  return i;
}

__device__ __forceinline__ void bspline3(float t, float (&w)[4]) {
  const float u = 1.f - t;
  w[0] = u * u * u * (1.f / 6.f);
  w[1] = (3.f * t * t * t - 6.f * t * t + 4.f) * (1.f / 6.f);
  w[2] = (-3.f * t * t * t + 3.f * t * t + 3.f * t + 1.f) * (1.f / 6.f);
  w[3] = t * t * t * (1.f / 6.f);
}

// ---- interp_order 2, 4, 5: the same machinery with the other B-splines ----------------------------------
// Centred cardinal B-spline of order 2 / 4 / 5 at distance a >= 0 (closed piecewise polynomials; SciPy evaluates the
// same functions in double and sets the last weight to 1 - sum of the others: equal to float32 rounding).
__device__ __forceinline__ float bspline_basis(float a, int order) {
  if (order == 2) return a <= 0.5f ? 0.75f - a * a : (a <= 1.5f ? 0.5f * (1.5f - a) * (1.5f - a) : 0.f);
  if (order == 4) {
    if (a <= 0.5f) return (115.f / 192.f) + a * a * (-0.625f + 0.25f * a * a);
    if (a <= 1.5f) return (55.f / 96.f) + a * ((5.f / 24.f) + a * (-1.25f + a * ((5.f / 6.f) - a * (1.f / 6.f))));
    const float u = 2.5f - a;
    return a <= 2.5f ? u * u * u * u * (1.f / 24.f) : 0.f;
  }
  // order 5
  if (a <= 1.f) return 0.55f + a * a * (-0.5f + a * a * (0.25f - a * (1.f / 12.f)));
  if (a <= 2.f) return 0.425f + a * (0.625f + a * (-1.75f + a * (1.25f + a * (-0.375f + a * (1.f / 24.f)))));
  const float u = 3.f - a;
  return a <= 3.f ? u * u * u * u * u * (1.f / 120.f) : 0.f;
}
// first tap of the order + 1 taps along one axis for the coordinate X + f and their weights (ni_interpolation.c: odd
// orders start at floor(c) - order / 2, even ones at floor(c + 0.5) - order / 2); w[k] for k > order is not set
__device__ __forceinline__ int spline_taps(int X, float f, int order, float (&w)[6]) {
  const int up = ((order & 1) == 0 && f >= 0.5f) ? 1 : 0;  // even order: the nearest sample is the centre tap
  const float t = f - static_cast<float>(up) + static_cast<float>(order / 2);  // distance of the coordinate from tap 0
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (k <= order) w[k] = bspline_basis(fabsf(t - static_cast<float>(k)), order);
  return X + up - order / 2;
}

__device__ __forceinline__ float sample_precip_cubic(const float *coef, const float *p, int X, int Y,
                                                     float fx, float fy, int m, int n,
                                                     float minval, int sorder = 3) {
  const bool outside = X < 0 || Y < 0 || X > n - 1 || Y > m - 1 || (X == n - 1 && fx > 0.f) ||
                       (Y == m - 1 && fy > 0.f);
  if (outside) return __builtin_nanf("");
  const int x1 = (X + 1 > n - 1) ? max(n - 2, 0) : X + 1;
  const int y1 = (Y + 1 > m - 1) ? max(m - 2, 0) : Y + 1;
  const unsigned r0 = static_cast<unsigned>(__mul24(Y, n)), r1 = static_cast<unsigned>(__mul24(y1, n));
  const float v00 = ld(p, (r0 + X) << 2), v01 = ld(p, (r0 + x1) << 2);
  const float v10 = ld(p, (r1 + X) << 2), v11 = ld(p, (r1 + x1) << 2);
  const Weights w = make_weights(fx, fy);
  const float finite = blend(w, isfinite(v00) ? 1.f : 0.f, isfinite(v01) ? 1.f : 0.f,
                             isfinite(v10) ? 1.f : 0.f, isfinite(v11) ? 1.f : 0.f);
  if (finite < 0.5f) return __builtin_nanf("");
  const float above = blend(w, v00 > minval ? 1.f : 0.f, v01 > minval ? 1.f : 0.f,
                            v10 > minval ? 1.f : 0.f, v11 > minval ? 1.f : 0.f);
  if (above < 0.5f) return minval;
  if (sorder != 3) {  // (uniform) orders 2, 4, 5: (order + 1)^2 taps, mirrored indices
    float gx[6], gy[6];
    const int x0 = spline_taps(X, fx, sorder, gx), y0 = spline_taps(Y, fy, sorder, gy);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (a > sorder) break;
      const unsigned row = static_cast<unsigned>(__mul24(mirror101(y0 + a, m), n));
      float line = 0.f;
#pragma unroll
      for (int b = 0; b < 6; ++b)
        if (b <= sorder) line = fmaf(gx[b], ld(coef, (row + mirror101(x0 + b, n)) << 2), line);
      acc = fmaf(gy[a], line, acc);
    }
    return acc;
  }
  float wx[4], wy[4];
  bspline3(fx, wx);
  bspline3(fy, wy);
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const unsigned row = static_cast<unsigned>(__mul24(mirror101(Y - 1 + a, m), n));
    float line = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) line = fmaf(wx[b], ld(coef, (row + mirror101(X - 1 + b, n)) << 2), line);
    acc = fmaf(wy[a], line, acc);
  }
  return acc;
}

// interp_order = 3 with a map_coordinates mode other than "constant" (oracle/semilag.py::_cubic_mode,
// pinned against SciPy): the coordinate is folded on the field's own lengths like for the lower
// orders; the two order-1 mask warps take the field's taps folded index by index (cval 0 for
// "grid-constant"); the 4 x 4 spline taps are folded index by index on the PADDED coefficient plane
// ((m + 2 npad, n + 2 npad): SciPy pads "nearest" and "grid-constant" by 12 samples before filtering
// and shifts the coordinate), "grid-constant" taps outside it are cval.  coef == nullptr: a non-finite
// cval was padded in and the filter's recursion carried it everywhere - every coefficient is NaN.
__device__ __forceinline__ float sample_precip_cubic_mode(const float *coef, const float *p, int X, int Y, float fx,
                                                          float fy, int m, int n, float minval, float cval, int mode,
                                                          int npad, int sorder = 3) {
  fold_coord(X, fx, n, mode);
  fold_coord(Y, fy, m, mode);
  {
    bool cx0 = false, cx1 = false, cy0 = false, cy1 = false;
    const int x0 = fold_tap(X, n, mode, &cx0), x1 = fold_tap(X + 1, n, mode, &cx1);
    const int y0 = fold_tap(Y, m, mode, &cy0), y1 = fold_tap(Y + 1, m, mode, &cy1);
    const unsigned r0 = static_cast<unsigned>(__mul24(y0, n)), r1 = static_cast<unsigned>(__mul24(y1, n));
    const float v00 = ld(p, (r0 + x0) << 2), v01 = ld(p, (r0 + x1) << 2);
    const float v10 = ld(p, (r1 + x0) << 2), v11 = ld(p, (r1 + x1) << 2);
    const bool k00 = !(cx0 || cy0), k01 = !(cx1 || cy0), k10 = !(cx0 || cy1), k11 = !(cx1 || cy1);  // not a cval tap
    const Weights w = make_weights(fx, fy);
    const float finite = blend(w, (k00 && isfinite(v00)) ? 1.f : 0.f, (k01 && isfinite(v01)) ? 1.f : 0.f,
                               (k10 && isfinite(v10)) ? 1.f : 0.f, (k11 && isfinite(v11)) ? 1.f : 0.f);
    if (finite < 0.5f) return __builtin_nanf("");
    const float above = blend(w, (k00 && v00 > minval) ? 1.f : 0.f, (k01 && v01 > minval) ? 1.f : 0.f,
                              (k10 && v10 > minval) ? 1.f : 0.f, (k11 && v11 > minval) ? 1.f : 0.f);
    if (above < 0.5f) return minval;
  }
  if (coef == nullptr) return __builtin_nanf("");
  const int big_m = m + 2 * npad, big_n = n + 2 * npad;
  if (sorder != 3) {  // (uniform) orders 2, 4, 5
    float gx[6], gy[6];
    const int x0 = spline_taps(X + npad, fx, sorder, gx), y0 = spline_taps(Y + npad, fy, sorder, gy);
    int gi[6];
    bool gc[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      gc[b] = false;
      gi[b] = b <= sorder ? fold_tap(x0 + b, big_n, mode, &gc[b]) : 0;
    }
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      if (a > sorder) break;
      bool yc = false;
      const unsigned row = static_cast<unsigned>(__mul24(fold_tap(y0 + a, big_m, mode, &yc), big_n));
      float line = 0.f;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        if (b <= sorder) {
          const float v = ld(coef, (row + gi[b]) << 2);
          line = fmaf(gx[b], (yc || gc[b]) ? cval : v, line);
        }
      }
      acc = fmaf(gy[a], line, acc);
    }
    return acc;
  }
  float wx[4], wy[4];
  bspline3(fx, wx);
  bspline3(fy, wy);
  int xi[4];
  bool xc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    xc[b] = false;
    xi[b] = fold_tap(X + npad - 1 + b, big_n, mode, &xc[b]);
  }
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    bool yc = false;
    const unsigned row = static_cast<unsigned>(__mul24(fold_tap(Y + npad - 1 + a, big_m, mode, &yc), big_n));
    float line = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float v = ld(coef, (row + xi[b]) << 2);
      line = fmaf(wx[b], (yc || xc[b]) ? cval : v, line);
    }
    acc = fmaf(wy[a], line, acc);
  }
  return acc;
}

}  // namespace sl
}  // namespace psh
