// Fused semi-Lagrangian backward advection for gfx950 (MI355X).
//
// Replaces the trajectory loop of pysteps/extrapolation/semilagrangian.py:200-255
// (extrapolate) together with its inner interpolate_motion (:181-198) and the
// order-0/1 scipy.ndimage.map_coordinates resampling it calls (:185-190 with
// mode="nearest", :225-232 with mode="constant").
//
// Design (see DESIGN.md "sl_fused"):
//  * A pixel's trajectory depends only on gathers from the constant velocity
//    field, so one thread owns one pixel for ALL T lead steps and keeps the
//    displacement D and the increment Vi in registers.  Nothing but the T output
//    planes (and optionally the final D) is ever written: algorithmic traffic
//    is 16*n_iter + 8 bytes per pixel per lead step.
//  * D is carried as integer + fraction (frac in [0,1)) per axis.  Sub-pixel
//    weights therefore keep full fp32 precision however far the trajectory has
//    travelled, and the "advected from outside" test of map_coordinates
//    (coord < 0 or coord > len-1, strict) becomes an integer comparison.
//  * 64x4-pixel workgroups: a wave reads 64 consecutive floats per tap row
//    (coalesced up to the sub-row shift), vertically adjacent waves share tap
//    rows through L1/L2.  The block index is remapped so that each XCD (block b
//    runs on XCD b % 8) owns one contiguous horizontal band of the image and
//    its private 4 MiB L2 sees all the halo reuse of that band.
//  * No LDS, no MFMA: the gather footprint moves with D and there is no dense
//    contraction.  The kernel is bound by HBM/LLC bandwidth.
#include "common.h"

// No implicit FMA contraction in this file: floor(w) and (w - floor(w)) must see
// the SAME rounded product w = sample * scale, otherwise a fused w - floor(w)
// disagrees with the integer part by one ulp of 1.0 and a trajectory that lands
// exactly on the domain edge is classified as outside.  FMAs are written out.
#pragma clang fp contract(off)

namespace psh {
namespace {

constexpr int kTileX = 64;
constexpr int kTileY = 4;

__device__ __forceinline__ float ld(const float *base, unsigned byte_off) {
  // uniform base + 32-bit lane offset -> global_load_dword v, v_off, s[base:base+1]
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}

// (i + f) -= w with f kept in [0,1); w is split exactly so |rounding| ~ 6e-8 px
__device__ __forceinline__ void retreat(int &i, float &f, float w) {
  const float wf = floorf(w);
  i -= static_cast<int>(wf);
  f -= (w - wf);
  if (f < 0.f) {
    f += 1.f;
    i -= 1;
  }
  if (f >= 1.f) {  // -1e-9 + 1 rounds up to 1.0f
    f = 0.f;
    i += 1;
  }
}

struct Taps {
  unsigned o00, o01, o10, o11;  // byte offsets inside one plane
  float w00, w01, w10, w11;
};

__device__ __forceinline__ void weights(Taps &t, float fx, float fy) {
  const float gx = 1.f - fx, gy = 1.f - fy;
  t.w00 = gy * gx;
  t.w01 = gy * fx;
  t.w10 = fy * gx;
  t.w11 = fy * fx;
}

__device__ __forceinline__ float blend(const Taps &t, float a, float b, float c, float d) {
  // all four products are formed: a NaN tap poisons the sample even at weight 0,
  // exactly like map_coordinates
  return fmaf(t.w11, d, fmaf(t.w10, c, fmaf(t.w01, b, t.w00 * a)));
}

// velocity taps, mode="nearest": the coordinate is clamped to [0,len-1]; with
// clamped indices both taps coincide outside the range, which is the same value.
__device__ __forceinline__ Taps vel_taps(int X, int Y, float fx, float fy, int m, int n) {
  Taps t;
  const int x0 = min(max(X, 0), n - 1), x1 = min(max(X + 1, 0), n - 1);
  const int y0 = min(max(Y, 0), m - 1), y1 = min(max(Y + 1, 0), m - 1);
  const unsigned r0 = static_cast<unsigned>(y0) * n, r1 = static_cast<unsigned>(y1) * n;
  t.o00 = (r0 + x0) * 4u;
  t.o01 = (r0 + x1) * 4u;
  t.o10 = (r1 + x0) * 4u;
  t.o11 = (r1 + x1) * 4u;
  weights(t, fx, fy);
  return t;
}

__device__ __forceinline__ bool interior(int X, int Y, int m, int n) {
  return static_cast<unsigned>(X) < static_cast<unsigned>(n - 1) &&
         static_cast<unsigned>(Y) < static_cast<unsigned>(m - 1);
}

__device__ __forceinline__ Taps interior_taps(int X, int Y, float fx, float fy, int n) {
  Taps t;
  t.o00 = (static_cast<unsigned>(Y) * n + X) * 4u;
  t.o01 = t.o00 + 4u;
  t.o10 = t.o00 + static_cast<unsigned>(n) * 4u;
  t.o11 = t.o10 + 4u;
  weights(t, fx, fy);
  return t;
}

__device__ __forceinline__ void sample_velocity(const float *u, const float *v, const Taps &t,
                                                float &su, float &sv) {
  const float a = ld(u, t.o00), b = ld(u, t.o01), c = ld(u, t.o10), d = ld(u, t.o11);
  const float e = ld(v, t.o00), f = ld(v, t.o01), g = ld(v, t.o10), h = ld(v, t.o11);
  su = blend(t, a, b, c, d);
  sv = blend(t, e, f, g, h);
}

// precip sample, mode="constant": outside -> outval; the upper tap at
// floor+1 == len is index-mirrored (weight 0 there).
template <int ORDER>
__device__ __forceinline__ float sample_precip(const float *p, int X, int Y, float fx, float fy,
                                               int m, int n, float outval) {
  const bool outside = X < 0 || Y < 0 || X > n - 1 || Y > m - 1 || (X == n - 1 && fx > 0.f) ||
                       (Y == m - 1 && fy > 0.f);
  if (outside) return outval;
  if (ORDER == 0) {
    // floor(c + 0.5): half rounds up
    const int xi = min(X + (fx >= 0.5f ? 1 : 0), n - 1);
    const int yi = min(Y + (fy >= 0.5f ? 1 : 0), m - 1);
    return ld(p, (static_cast<unsigned>(yi) * n + xi) * 4u);
  }
  Taps t;
  const int x1 = (X + 1 > n - 1) ? max(n - 2, 0) : X + 1;
  const int y1 = (Y + 1 > m - 1) ? max(m - 2, 0) : Y + 1;
  const unsigned r0 = static_cast<unsigned>(Y) * n, r1 = static_cast<unsigned>(y1) * n;
  t.o00 = (r0 + X) * 4u;
  t.o01 = (r0 + x1) * 4u;
  t.o10 = (r1 + X) * 4u;
  t.o11 = (r1 + x1) * 4u;
  weights(t, fx, fy);
  return blend(t, ld(p, t.o00), ld(p, t.o01), ld(p, t.o10), ld(p, t.o11));
}

template <int ORDER, bool HAS_PRECIP>
__global__ __launch_bounds__(kTileX *kTileY) void semilag_fused(SemilagArgs a, int tiles_x,
                                                                int n_tiles, int tiles_per_xcd) {
  // XCD-aware remap: hardware block b -> XCD b % 8; give XCD k the k-th band of tiles
  const int b = blockIdx.x;
  const int tile = (b % kNumXcd) * tiles_per_xcd + b / kNumXcd;
  if (tile >= n_tiles) return;
  const int x = (tile % tiles_x) * kTileX + (threadIdx.x & (kTileX - 1));
  const int y = (tile / tiles_x) * kTileY + (threadIdx.x / kTileX);
  const int m = a.m, n = a.n;
  if (x >= n || y >= m) return;

  const size_t plane = static_cast<size_t>(m) * n;
  const float *__restrict__ u = a.vel;
  const float *__restrict__ v = a.vel + plane;
  const unsigned pix = (static_cast<unsigned>(y) * n + x) * 4u;
  const float sub = a.n_iter > 1 ? static_cast<float>(a.n_iter) : 1.f;

  int dix = 0, diy = 0;
  float dfx = 0.f, dfy = 0.f, vix, viy;

  auto motion_at = [&](int X, int Y, float fx, float fy, float s) {
    float su, sv;
    if (interior(X, Y, m, n)) {
      sample_velocity(u, v, interior_taps(X, Y, fx, fy, n), su, sv);
    } else {
      sample_velocity(u, v, vel_taps(X, Y, fx, fy, m, n), su, sv);
    }
    if (sub != 1.f) {
      su /= sub;
      sv /= sub;
    }
    vix = su * s;
    viy = sv * s;
  };

  const float s0 = a.scale[0];
  if (a.resume) {
    const double px = a.disp[static_cast<size_t>(y) * n + x];
    const double py = a.disp[plane + static_cast<size_t>(y) * n + x];
    const double flx = floor(px), fly = floor(py);
    dix = static_cast<int>(flx);
    diy = static_cast<int>(fly);
    dfx = static_cast<float>(px - flx);
    dfy = static_cast<float>(py - fly);
    if (dfx >= 1.f) {  // fraction rounded up to 1.0f
      dfx = 0.f;
      dix += 1;
    }
    if (dfy >= 1.f) {
      dfy = 0.f;
      diy += 1;
    }
    motion_at(x + dix, y + diy, dfx, dfy, s0);
  } else {
    // first increment is NOT divided by n_iter (semilagrangian.py:202)
    vix = ld(u, pix) * s0;
    viy = ld(v, pix) * s0;
  }

  float *__restrict__ out = a.out;
  for (int t = 0; t < a.T; ++t) {
    const float s = a.scale[t];
    if (a.n_iter > 0) {
      for (int k = 0; k < a.n_iter; ++k) {
        int mx = dix, my = diy;
        float gx = dfx, gy = dfy;
        retreat(mx, gx, 0.5f * vix);
        retreat(my, gy, 0.5f * viy);
        motion_at(x + mx, y + my, gx, gy, s);  // midpoint rule (:213)
        retreat(dix, dfx, vix);
        retreat(diy, dfy, viy);
        motion_at(x + dix, y + diy, dfx, dfy, s);
      }
    } else {
      if (t > 0 || a.resume) motion_at(x + dix, y + diy, dfx, dfy, s);
      retreat(dix, dfx, vix);
      retreat(diy, dfy, viy);
    }
    if (HAS_PRECIP) {
      const float val = sample_precip<ORDER>(a.precip, x + dix, y + diy, dfx, dfy, m, n, a.outval);
      *reinterpret_cast<float *>(reinterpret_cast<char *>(out) + pix) = val;
      out += plane;
    }
  }

  if (a.disp != nullptr) {
    a.disp[static_cast<size_t>(y) * n + x] = static_cast<double>(dix) + static_cast<double>(dfx);
    a.disp[plane + static_cast<size_t>(y) * n + x] =
        static_cast<double>(diy) + static_cast<double>(dfy);
  }
}

}  // namespace

hipError_t launch_semilag(const SemilagArgs &a, hipStream_t stream) {
  const int tiles_x = (a.n + kTileX - 1) / kTileX;
  const int tiles_y = (a.m + kTileY - 1) / kTileY;
  const int n_tiles = tiles_x * tiles_y;
  const int tiles_per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
  const dim3 grid(tiles_per_xcd * kNumXcd), block(kTileX * kTileY);
  const bool has_precip = a.precip != nullptr;
  if (!has_precip) {
    hipLaunchKernelGGL((semilag_fused<1, false>), grid, block, 0, stream, a, tiles_x, n_tiles,
                       tiles_per_xcd);
  } else if (a.order == 0) {
    hipLaunchKernelGGL((semilag_fused<0, true>), grid, block, 0, stream, a, tiles_x, n_tiles,
                       tiles_per_xcd);
  } else {
    hipLaunchKernelGGL((semilag_fused<1, true>), grid, block, 0, stream, a, tiles_x, n_tiles,
                       tiles_per_xcd);
  }
  return hipGetLastError();
}

}  // namespace psh
