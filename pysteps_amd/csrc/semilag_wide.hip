// Semi-Lagrangian backward advection, three pixels per lane (gfx950 / MI355X).
//
// Same algorithm and the same arithmetic, operation for operation, as semilag_fused in
// semilag.hip (pysteps/extrapolation/semilagrangian.py:181-255, interp_order 0/1); what changes
// is how the taps reach the registers.  Measured on MI355X (tools/gather_probe.py, DESIGN.md
// 3.1): a wave64 buffer load costs the CU's vector memory pipeline ~8 clocks as dword, ~16 as
// dwordx2 and ~16 as dwordx4 - whatever the alignment and however few lanes are active - so
// only 16-byte loads reach the 64 B/clk of the L1.
//
// Here a lane owns THREE horizontally adjacent pixels.  In a smooth motion field their
// trajectories stay rigid - same row, consecutive columns - and one dwordx4 load per tap row
// and plane brings columns X..X+3, i.e. all the taps of the three pixels: no neighbour
// exchange, no extra loads.  Where an integer boundary of the trajectory falls inside a lane's
// run (the run splits in two rigid pieces) the wave issues a second set of loads at the second
// piece's anchor and every pixel blends from the set it belongs to; a third piece (pixel 2 on
// neither) and lanes that touch the border take clamped scalar gathers; those are rare and
// exec-masked, a wave without any skips that code.
//
// Opt-in (psh_set_option("semilag_variant", 3)), measured at 4096^2 x 24, n_iter 1: 1.24 ms
// against 1.45 ms for semilag_fused in uniform motion, but 1.78 against 1.59 ms in the sheared
// test field, where after a few lead steps nearly every 192-pixel wave carries a split and
// pays the second set of loads and blends.
//
// Work decomposition: 192 x 4 pixel workgroups, one image row per wave; block index
// remapped so that each XCD owns one contiguous band of tiles (common.h kNumXcd).
#include "common.h"

// see semilag.hip: floor(t) and t - floor(t) must see the same rounded t
#pragma clang fp contract(off)

#include "semilag_device.h"

namespace psh {
namespace {

using namespace sl;

constexpr int kQ = 3;            // pixels per lane, along x
constexpr int kWideTileX = 64 * kQ;
constexpr int kWideRows = 4;     // waves per workgroup, one image row each

typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint3v __attribute__((ext_vector_type(3)));

enum : int { kVel = 1, kPrecip = 2 };

struct Planes {
  __amdgpu_buffer_rsrc_t ru, rv, rp;  // u, v and the advected field as raw buffers
  const float *u, *v, *p;
  int row_bytes;
};

__device__ __forceinline__ float4v bld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soff) {
  return __builtin_bit_cast(float4v,
                            __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(byte_off), soff, 0));
}

// velocity, mode="nearest" (clamped indices), any position
__device__ __forceinline__ void velocity_clamped(const Planes &F, int X, int Y, float fx, float fy, int m,
                                                 int n, float &su, float &sv) {
  const int x0 = min(max(X, 0), n - 1), x1 = min(max(X + 1, 0), n - 1);
  const int y0 = min(max(Y, 0), m - 1), y1 = min(max(Y + 1, 0), m - 1);
  const unsigned r0 = static_cast<unsigned>(__mul24(y0, n)), r1 = static_cast<unsigned>(__mul24(y1, n));
  const unsigned o00 = (r0 + x0) << 2, o01 = (r0 + x1) << 2, o10 = (r1 + x0) << 2, o11 = (r1 + x1) << 2;
  const float a = ld(F.u, o00), b = ld(F.u, o01), c = ld(F.u, o10), d = ld(F.u, o11);
  const float e = ld(F.v, o00), f = ld(F.v, o01), g = ld(F.v, o10), h = ld(F.v, o11);
  const Weights w = make_weights(fx, fy);
  su = blend(w, a, b, c, d);
  sv = blend(w, e, f, g, h);
}

// The four columns c[0..3] of two tap rows of one plane
struct Rows {
  float4v r0, r1;
};

template <int ORDER>
__device__ __forceinline__ float precip_from_rows(const Rows &P, int i, float fx, float fy) {
  if (ORDER == 1) return blend(make_weights(fx, fy), P.r0[i], P.r0[i + 1], P.r1[i], P.r1[i + 1]);
  // nearest neighbour, half rounds up (map_coordinates order 0)
  const float top = fx >= 0.5f ? P.r0[i + 1] : P.r0[i], bot = fx >= 0.5f ? P.r1[i + 1] : P.r1[i];
  return fy >= 0.5f ? bot : top;
}

// Sample velocity and / or the advected field at the three positions of a lane.
template <int ORDER, int WHAT>
__device__ __forceinline__ void sample_wide(const Planes &F, const int (&X)[kQ], const int (&Y)[kQ],
                                            const float (&fx)[kQ], const float (&fy)[kQ], int m, int n,
                                            float outval, float (&su)[kQ], float (&sv)[kQ],
                                            float (&sp)[kQ]) {
  constexpr bool kV = (WHAT & kVel) != 0, kP = (WHAT & kPrecip) != 0;
  // anchor A: pixel 0.  Pixel i belongs to it when it sits i columns to the right in the same row.
  const int ax = X[0], ay = Y[0];
  const bool on_a1 = X[1] - 1 == ax && Y[1] == ay, on_a2 = X[2] - 2 == ax && Y[2] == ay;
  // anchor B: the first pixel that left A, shifted back to where its pixel 0 would be
  const int bx = on_a1 ? X[2] - 2 : X[1] - 1, by = on_a1 ? Y[2] : Y[1];
  const bool need_b = !(on_a1 && on_a2);
  const bool on_b2 = X[2] - 2 == bx && Y[2] == by;  // (pixel 1 is on A or defines B)
  const bool covered = on_a2 || on_b2;
  // columns anchor..anchor+3 and rows anchor, anchor+1 strictly inside the image
  const bool a_in = static_cast<unsigned>(ax) < static_cast<unsigned>(n - 3) &&
                    static_cast<unsigned>(ay) < static_cast<unsigned>(m - 1);
  const bool b_in = static_cast<unsigned>(bx) < static_cast<unsigned>(n - 3) &&
                    static_cast<unsigned>(by) < static_cast<unsigned>(m - 1);
  // pixels 0 and 1 are on A or B by construction; pixel 2 may be on neither (a second split
  // inside the run): it alone then takes the scalar gathers
  const bool fast = a_in && (!need_b || b_in);
  const bool stray2 = fast && !covered;
  const unsigned long long fast_mask = __builtin_amdgcn_ballot_w64(fast);
  const bool wave_needs_b = __builtin_amdgcn_ballot_w64(fast && need_b) != 0;
  // lanes outside the fast path load from the image origin (in bounds, ignored): no exec
  // juggling around the wide loads
  const unsigned off_a = fast ? static_cast<unsigned>(__mul24(ay, n) + ax) << 2 : 0u;
  const int rb = F.row_bytes;

  Rows ua, va, pa;
  if (kV) {
    ua.r0 = bld4(F.ru, off_a, 0);
    ua.r1 = bld4(F.ru, off_a, rb);
    va.r0 = bld4(F.rv, off_a, 0);
    va.r1 = bld4(F.rv, off_a, rb);
  }
  if (kP) {
    pa.r0 = bld4(F.rp, off_a, 0);
    pa.r1 = bld4(F.rp, off_a, rb);
  }
  if (!wave_needs_b) {
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      if (kV) {
        const Weights w = make_weights(fx[i], fy[i]);
        su[i] = blend(w, ua.r0[i], ua.r0[i + 1], ua.r1[i], ua.r1[i + 1]);
        sv[i] = blend(w, va.r0[i], va.r0[i + 1], va.r1[i], va.r1[i + 1]);
      }
      if (kP) sp[i] = precip_from_rows<ORDER>(pa, i, fx[i], fy[i]);
    }
  } else {
    // second set at anchor B (lanes that do not need it read A's again: L1 hits)
    const unsigned off_b = (fast && need_b) ? static_cast<unsigned>(__mul24(by, n) + bx) << 2 : off_a;
    Rows ub, vb, pb;
    if (kV) {
      ub.r0 = bld4(F.ru, off_b, 0);
      ub.r1 = bld4(F.ru, off_b, rb);
      vb.r0 = bld4(F.rv, off_b, 0);
      vb.r1 = bld4(F.rv, off_b, rb);
    }
    if (kP) {
      pb.r0 = bld4(F.rp, off_b, 0);
      pb.r1 = bld4(F.rp, off_b, rb);
    }
    const bool from_a[kQ] = {true, on_a1, on_a2};
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      if (kV) {
        const Weights w = make_weights(fx[i], fy[i]);
        su[i] = blend(w, ua.r0[i], ua.r0[i + 1], ua.r1[i], ua.r1[i + 1]);
        sv[i] = blend(w, va.r0[i], va.r0[i + 1], va.r1[i], va.r1[i + 1]);
        if (i > 0) {
          const float tu = blend(w, ub.r0[i], ub.r0[i + 1], ub.r1[i], ub.r1[i + 1]);
          const float tv = blend(w, vb.r0[i], vb.r0[i + 1], vb.r1[i], vb.r1[i + 1]);
          su[i] = from_a[i] ? su[i] : tu;
          sv[i] = from_a[i] ? sv[i] : tv;
        }
      }
      if (kP) {
        sp[i] = precip_from_rows<ORDER>(pa, i, fx[i], fy[i]);
        if (i > 0) {
          const float tp = precip_from_rows<ORDER>(pb, i, fx[i], fy[i]);
          sp[i] = from_a[i] ? sp[i] : tp;
        }
      }
    }
  }
  // ---- pixels off the fast path: clamped gathers, pixel by pixel ---------------------------
  if (__builtin_amdgcn_ballot_w64(stray2) != 0) {
    if (stray2) {
      if (kV) velocity_clamped(F, X[2], Y[2], fx[2], fy[2], m, n, su[2], sv[2]);
      if (kP) sp[2] = sample_precip_border<ORDER>(F.p, X[2], Y[2], fx[2], fy[2], m, n, outval);
    }
  }
  if (fast_mask != __builtin_amdgcn_ballot_w64(true)) {
    if (!fast) {
#pragma unroll
      for (int i = 0; i < kQ; ++i) {
        if (kV) velocity_clamped(F, X[i], Y[i], fx[i], fy[i], m, n, su[i], sv[i]);
        if (kP) sp[i] = sample_precip_border<ORDER>(F.p, X[i], Y[i], fx[i], fy[i], m, n, outval);
      }
    }
  }
}

template <int ORDER, bool HAS_PRECIP>
__global__ __launch_bounds__(64 * kWideRows) void semilag_wide(
    const float *__restrict__ precip, const float *__restrict__ vel, float *__restrict__ out,
    double *__restrict__ disp, const float *__restrict__ scale, float first_scale, int m, int n, int T,
    int n_iter, int resume, float outval, int row0, int rows, int tiles_x, int n_tiles,
    int tiles_per_xcd) {
  const int blk = blockIdx.x;
  const int tile = (blk % kNumXcd) * tiles_per_xcd + blk / kNumXcd;
  if (tile >= n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xt0 = (tile % tiles_x) * kWideTileX + kQ * lane;
  const int yt = row0 + (tile / tiles_x) * kWideRows + wave;
  // lanes / rows past the edge shadow the edge pixel; only their stores are masked
  const int y = min(yt, m - 1);
  const bool row_live = yt < row0 + rows;
  const size_t plane = static_cast<size_t>(m) * n;
  const int plane_bytes = static_cast<int>(plane * sizeof(float));
  Planes F;
  F.u = vel;
  F.v = vel + plane;
  F.p = precip;
  F.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel), 0, plane_bytes, 0x00020000);
  F.rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel + plane), 0, plane_bytes, 0x00020000);
  F.rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HAS_PRECIP ? precip : vel), 0, plane_bytes,
                                           0x00020000);
  F.row_bytes = n * static_cast<int>(sizeof(float));

  int xs[kQ], px[kQ], py[kQ];
  float fx[kQ], fy[kQ], vix[kQ], viy[kQ], su[kQ], sv[kQ], sp[kQ];
  bool live[kQ];
#pragma unroll
  for (int i = 0; i < kQ; ++i) {
    live[i] = row_live && xt0 + i < n;
    xs[i] = min(xt0 + i, n - 1);
    px[i] = xs[i];
    py[i] = y;
    fx[i] = fy[i] = sp[i] = 0.f;
  }
  const size_t row_elems = static_cast<size_t>(y) * n;

  if (resume) {
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      const double dx = disp[row_elems + xs[i]];
      const double dy = disp[plane + row_elems + xs[i]];
      const double flx = floor(dx), fly = floor(dy);
      px[i] += static_cast<int>(flx);
      py[i] += static_cast<int>(fly);
      fx[i] = fminf(static_cast<float>(dx - flx), kMaxFrac);
      fy[i] = fminf(static_cast<float>(dy - fly), kMaxFrac);
    }
    sample_wide<ORDER, kVel>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
    const float s0 = scale[0];
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      vix[i] = su[i] * s0;
      viy[i] = sv[i] * s0;
    }
  } else {
    // first increment is NOT divided by n_iter (semilagrangian.py:202)
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      const unsigned pix = static_cast<unsigned>(__mul24(y, n) + xs[i]) << 2;
      vix[i] = ld(F.u, pix) * first_scale;
      viy[i] = ld(F.v, pix) * first_scale;
    }
  }
  // with n_iter > 0 the increment is only ever used halved (midpoint rule): carry Vi / 2
  if (n_iter > 0) {
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      vix[i] *= 0.5f;
      viy[i] *= 0.5f;
    }
  }

  // band-local output row of this wave
  float *orow = out + static_cast<size_t>(y - row0) * n;
  for (int t = 0; t < T; ++t) {
    const float s = scale[t];  // (lead-time increment / vel_timestep) / max(n_iter, 1)
    if (n_iter > 0) {
      const float half_s = 0.5f * s;
      for (int k = 0; k < n_iter; ++k) {
        int mx[kQ], my[kQ];
        float gx[kQ], gy[kQ];
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
          mx[i] = px[i];
          my[i] = py[i];
          gx[i] = fx[i];
          gy[i] = fy[i];
          retreat(mx[i], gx[i], vix[i]);  // midpoint rule (:213), vix = Vi / 2
          retreat(my[i], gy[i], viy[i]);
        }
        sample_wide<ORDER, kVel>(F, mx, my, gx, gy, m, n, outval, su, sv, sp);
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
          retreat(px[i], fx[i], su[i] * s);
          retreat(py[i], fy[i], sv[i] * s);
        }
        if (HAS_PRECIP && k == n_iter - 1) {
          sample_wide<ORDER, kVel | kPrecip>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
        } else {
          sample_wide<ORDER, kVel>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
        }
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
          vix[i] = su[i] * half_s;
          viy[i] = sv[i] * half_s;
        }
      }
    } else {
      if (t > 0 || resume) {
        sample_wide<ORDER, kVel>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
#pragma unroll
        for (int i = 0; i < kQ; ++i) {
          vix[i] = su[i] * s;
          viy[i] = sv[i] * s;
        }
      }
#pragma unroll
      for (int i = 0; i < kQ; ++i) {
        retreat(px[i], fx[i], vix[i]);
        retreat(py[i], fy[i], viy[i]);
      }
      if (HAS_PRECIP) sample_wide<ORDER, kPrecip>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
    }
    if (HAS_PRECIP) {
      // streamed once, never re-read: nontemporal, one 12-byte store per lane where the whole
      // run is inside (consecutive lanes then write one contiguous 768-byte stretch)
      if (live[kQ - 1]) {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(orow, 0, F.row_bytes, 0x00020000);
        const uint3v q = {__float_as_uint(sp[0]), __float_as_uint(sp[1]), __float_as_uint(sp[2])};
        __builtin_amdgcn_raw_buffer_store_b96(q, ro, xt0 * 4, 0, 2 /* nt */);
      } else {
#pragma unroll
        for (int i = 0; i < kQ - 1; ++i)
          if (live[i]) __builtin_nontemporal_store(sp[i], orow + xt0 + i);
      }
      orow += static_cast<size_t>(rows) * n;
    }
  }

  if (disp != nullptr) {
#pragma unroll
    for (int i = 0; i < kQ; ++i) {
      if (!live[i]) continue;
      disp[row_elems + xs[i]] = static_cast<double>(px[i] - xs[i]) + static_cast<double>(fx[i]);
      disp[plane + row_elems + xs[i]] = static_cast<double>(py[i] - y) + static_cast<double>(fy[i]);
    }
  }
}

}  // namespace

// interp_order 0 / 1 on images at least one tile wide; everything else stays with semilag_fused
bool semilag_wide_eligible(const SemilagArgs &a) {
  return a.order != 3 && a.bmode == 0 && a.n >= kWideTileX && a.m >= 2;
}

hipError_t launch_semilag_wide(const SemilagArgs &a, hipStream_t stream) {
  const int tiles_x = (a.n + kWideTileX - 1) / kWideTileX;
  const int tiles_y = (a.rows + kWideRows - 1) / kWideRows;
  const int n_tiles = tiles_x * tiles_y;
  const int tiles_per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
  const dim3 grid(tiles_per_xcd * kNumXcd), block(64 * kWideRows);
#define PSH_SLW_LAUNCH(ORDER, HASP)                                                                 \
  hipLaunchKernelGGL((semilag_wide<ORDER, HASP>), grid, block, 0, stream, a.precip, a.vel, a.out,   \
                     a.disp, a.scale, a.first_scale, a.m, a.n, a.T, a.n_iter, a.resume, a.outval,   \
                     a.row0, a.rows, tiles_x, n_tiles, tiles_per_xcd)
  if (a.precip == nullptr) {
    PSH_SLW_LAUNCH(1, false);
  } else if (a.order == 0) {
    PSH_SLW_LAUNCH(0, true);
  } else {
    PSH_SLW_LAUNCH(1, true);
  }
#undef PSH_SLW_LAUNCH
  return hipGetLastError();
}

}  // namespace psh
