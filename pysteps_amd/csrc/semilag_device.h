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
__device__ __forceinline__ void retreat(int &P, float &f, float w) {
  const float t = f - w;
  const float k = floorf(t);
  P += static_cast<int>(k);
  f = fminf(t - k, kMaxFrac);  // (-1e-9) - (-1) rounds to 1.0f: keep f < 1
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


}  // namespace sl
}  // namespace psh
