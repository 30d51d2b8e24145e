// Fused semi-Lagrangian backward advection for gfx950 (MI355X).
//
// Replaces the trajectory loop of pysteps/extrapolation/semilagrangian.py:200-255
// (extrapolate) together with its inner interpolate_motion (:181-198) and the
// order-0/1 scipy.ndimage.map_coordinates resampling it calls (:185-190 with
// mode="nearest", :225-232 with mode="constant").
//
// Design of the gather kernels (their history: docs/history.md 3.1) - the window kernel, the default
// since round 5, is described where it starts below and in DESIGN.md 3.1:
//  * A pixel's trajectory depends only on gathers from the constant velocity
//    field, so one thread owns one pixel for ALL T lead steps and keeps the
//    displacement D and the increment Vi in registers.  Nothing but the T output
//    planes (and optionally the final D) is ever written: algorithmic traffic
//    is 16*n_iter + 8 bytes per pixel per lead step.
//  * The trajectory is carried as integer pixel position + fraction in [0,1) per axis.  Sub-pixel
//    weights therefore keep full fp32 precision however far the trajectory has
//    travelled, and the "advected from outside" test of map_coordinates
//    (coord < 0 or coord > len-1, strict) becomes an integer comparison.
//  * 64x8-pixel workgroups of 8 waves, one image row per wave, one pixel per lane (two rows
//    per thread were measured equal).  A wave reads 64 consecutive floats per tap row
//    (coalesced up to the sub-row shift).  The block index is remapped so that each XCD
//    (block b runs on XCD b % 8) owns one contiguous horizontal band of the image and its
//    private 4 MiB L2 sees all the halo reuse of that band.
//  * Waves whose 64 lanes all have their four taps strictly inside the image (almost all of
//    them) take a clamp-free path: buffer loads with one lane offset for every plane (+1 row
//    = scalar offset), the right-hand column of each lane's 2x2 footprint taken from lane
//    i+1 by DPP (v_cndmask_b32_dpp) unless the neighbour's trajectory sits elsewhere.
//  * No LDS in the gather kernels, no MFMA anywhere: the gather footprint moves with D and there is no
//    dense contraction.  The kernel is bound by the CU's vector-memory pipeline (docs/history.md 3.1).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"

// No implicit FMA contraction in this file: floor(t) and (t - floor(t)) must see
// the SAME rounded value t = frac - sample * scale, otherwise a fused form
// disagrees with the integer part by one ulp of 1.0 and a trajectory that lands
// exactly on the domain edge is classified as outside.  FMAs are written out.
#pragma clang fp contract(off)

#include "semilag_device.h"

namespace psh {
namespace {

using namespace sl;

constexpr int kTileX = 64;
// kernel flavours: kModeDirect: one plane per component, DPP column sharing (short calls, no packed plane);
// kModePacked: {u,v} interleaved velocity plane; kModePacked2: that plus the row-pair field plane
enum : int { kModeDirect = 0, kModePacked = 3, kModePacked2 = 4 };
// the direct kernel runs 8 rows per workgroup: the tap row below a wave's pixels is the row the next
// wave samples, more rows per workgroup = more of that reuse in the CU's L1 (1.55 -> 1.50 ms)
constexpr int kDirectWaves = 8;
struct Fields {
  const float *u0, *v0, *p0;  // plane bases (border path, scalar loads)
  // buffer descriptors of the three planes for the fast path: addressing is then
  // descriptor (SGPR) + one 32-bit lane offset + immediate (+1 column) + scalar
  // offset (+1 row) - no per-load 64-bit VALU address arithmetic
  __amdgpu_buffer_rsrc_t ru, rv, rp;
  __amdgpu_buffer_rsrc_t ruv;  // packed {u,v} float2 plane (kModePacked, kModePacked2)
  __amdgpu_buffer_rsrc_t rpp;  // row-pair field plane {p(y,x), p(y+1,x)} (kModePacked2)
  int row_bytes;
  const float *coef;  // cubic B-spline coefficients of the field (interp_order 3 only)
  int cpad;           // ... padded by this many samples (boundary modes "nearest", "grid-constant")
  float minval;       // minimum over its finite values (interp_order 3 only)
  int sorder;         // ... and which B-spline the coefficients belong to: 2, 3, 4 or 5
  int bmode;          // boundary mode of the field resampling (semilag_device.h kMode*)
};

__device__ __forceinline__ float bld(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, static_cast<int>(byte_off), soff, 0));
}

// ---- cross-lane helpers ---------------------------------------------------------------
// lane i receives the value of lane i+1 (v_mov_b32_dpp wave_shl:1); lane 63, which has no
// right neighbour, reads 0 (bound_ctrl)
__device__ __forceinline__ unsigned from_next_lane_or_zero(unsigned v) {
  return static_cast<unsigned>(
      __builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// a VGPR whose content does not matter (lanes that are overwritten before use): spares the
// zero-initialisation the compiler would otherwise emit in front of exec-masked loads
__device__ __forceinline__ float any_value() {
  float v;
  asm volatile("" : "=v"(v));
  return v;
}

// right[i] = own[i] ? right[i] : left[i + 1] for two / three planes x two tap rows, one VALU
// instruction per value: v_cndmask_b32_dpp selects between the lane's own register and the
// DPP-shifted left column of its neighbour.  Lane 63 has no neighbour: its write is
// disabled by the DPP rule for invalid source lanes, and it is in `own` anyway.
// s_nop 1 covers the two wait states a DPP read needs after a VALU write of its source.
#define PSH_TAKE(R, L) "v_cndmask_b32_dpp %[" #R "], %[" #L "], %[" #R "], vcc wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void take_right_columns(unsigned long long own, float a, float c, float e,
                                                   float g, float &b, float &d, float &f, float &h) {
  asm("s_mov_b64 vcc, %[own]\n\ts_nop 1\n\t" PSH_TAKE(b, a) PSH_TAKE(d, c) PSH_TAKE(f, e) PSH_TAKE(h, g)
      : [b] "+v"(b), [d] "+v"(d), [f] "+v"(f), [h] "+v"(h)
      : [a] "v"(a), [c] "v"(c), [e] "v"(e), [g] "v"(g), [own] "s"(own)
      : "vcc");
}
__device__ __forceinline__ void take_right_columns(unsigned long long own, float a, float c, float &b,
                                                   float &d) {
  asm("s_mov_b64 vcc, %[own]\n\ts_nop 1\n\t" PSH_TAKE(b, a) PSH_TAKE(d, c)
      : [b] "+v"(b), [d] "+v"(d)
      : [a] "v"(a), [c] "v"(c), [own] "s"(own)
      : "vcc");
}
#undef PSH_TAKE

// ---- fast path: every lane of the wave has all four taps strictly inside ----
// The L1 (TCP) moves 64 B/clk/CU, so the fast path asks it for as few bytes as
// possible: each lane loads only the LEFT column of its 2x2 footprint (one lane
// offset serves all planes, the +1 row is a second uniform base) and takes the
// RIGHT column from lane i+1 by DPP.  That is valid wherever the neighbour's
// integer position is exactly one pixel to the right (the rule in a smooth motion
// field); the few other lanes - trajectory crossing an integer boundary, lane 63 -
// fetch their right column themselves in an exec-masked branch issued together
// with the main loads.  All NPX pixels of the thread are loaded before any is used.
template <int NPX, bool WITH_P>
__device__ __forceinline__ void sample_interior(const Fields &F, const int (&X)[NPX],
                                                const int (&Y)[NPX], const float (&fx)[NPX],
                                                const float (&fy)[NPX], int n, float (&su)[NPX],
                                                float (&sv)[NPX], float (&sp)[NPX]) {
  unsigned off[NPX];
  unsigned long long own[NPX];
  float a[NPX], c[NPX], e[NPX], g[NPX], pa[NPX], pc[NPX];
  float b[NPX], d[NPX], f[NPX], h[NPX], pb[NPX], pd[NPX];
  float lb[NPX], ld_[NPX], lf[NPX], lh[NPX], lpb[NPX], lpd[NPX];  // lane 63's right column (SGPRs)
  const int rb = F.row_bytes;
  const bool last_lane = (threadIdx.x & 63) == 63;
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    off[j] = static_cast<unsigned>(__mul24(Y[j], n) + X[j]) << 2;
    // Lane 63 has no right neighbour.  Its right column is one address per wave: fetched with
    // SCALAR loads (s_load_dword through the scalar cache), it costs the vector memory pipeline
    // nothing - an exec-masked vector load for one lane would cost it as much as a full one
    // (tools/gather_probe.py), and in smooth motion lane 63 is the only lane that needs one.
    const unsigned off63 = static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(off[j]), 63)) + 4u;
    const unsigned off63_below = off63 + static_cast<unsigned>(rb);
    lb[j] = ld(F.u0, off63);
    ld_[j] = ld(F.u0, off63_below);
    lf[j] = ld(F.v0, off63);
    lh[j] = ld(F.v0, off63_below);
    if (WITH_P) {
      lpb[j] = ld(F.p0, off63);
      lpd[j] = ld(F.p0, off63_below);
    }
    // interior positions have X + 1 <= n - 1, so "the neighbour's linear offset is mine + 1"
    // is the same statement as "same row, next column"
    const bool own_right = from_next_lane_or_zero(off[j]) != off[j] + 4u && !last_lane;
    own[j] = __builtin_amdgcn_ballot_w64(own_right) | (1ull << 63);
    b[j] = any_value(), d[j] = any_value(), f[j] = any_value(), h[j] = any_value();
    if (WITH_P) pb[j] = any_value(), pd[j] = any_value();
    if (own_right) {  // skipped by the whole wave (s_cbranch_execz) when no trajectory crossed
      b[j] = bld(F.ru, off[j] + 4u, 0);
      d[j] = bld(F.ru, off[j] + 4u, rb);
      f[j] = bld(F.rv, off[j] + 4u, 0);
      h[j] = bld(F.rv, off[j] + 4u, rb);
      if (WITH_P) {
        pb[j] = bld(F.rp, off[j] + 4u, 0);
        pd[j] = bld(F.rp, off[j] + 4u, rb);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    a[j] = bld(F.ru, off[j], 0);
    c[j] = bld(F.ru, off[j], rb);
    e[j] = bld(F.rv, off[j], 0);
    g[j] = bld(F.rv, off[j], rb);
    if (WITH_P) {
      pa[j] = bld(F.rp, off[j], 0);
      pc[j] = bld(F.rp, off[j], rb);
    }
  }
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    const Weights w = make_weights(fx[j], fy[j]);
    b[j] = last_lane ? lb[j] : b[j];
    d[j] = last_lane ? ld_[j] : d[j];
    f[j] = last_lane ? lf[j] : f[j];
    h[j] = last_lane ? lh[j] : h[j];
    take_right_columns(own[j], a[j], c[j], e[j], g[j], b[j], d[j], f[j], h[j]);
    su[j] = blend(w, a[j], b[j], c[j], d[j]);
    sv[j] = blend(w, e[j], f[j], g[j], h[j]);
    if (WITH_P) {
      pb[j] = last_lane ? lpb[j] : pb[j];
      pd[j] = last_lane ? lpd[j] : pd[j];
      take_right_columns(own[j], pa[j], pc[j], pb[j], pd[j]);
      sp[j] = blend(w, pa[j], pb[j], pc[j], pd[j]);
    }
  }
}

// ---- fast path over the packed velocity plane ---------------------------------------------
// What a gather costs the vector memory pipeline depends on the instruction, not on the bytes
// (tools/gather_probe.py): a wave64 dword load ~8-9.5 clk, dwordx2 and dwordx4 both ~16.5 clk,
// any alignment.  With the two velocity components interleaved ({u,v} float2 per pixel,
// pack_velocity below) ONE dwordx4 at the lane's own position returns u and v of BOTH columns of
// a tap row, so a velocity sampling pass is 2 loads and the field adds one dwordx2 per tap row
// ({p(X), p(X+1)}): 6 loads per pixel and lead step instead of 10 + 10 - and no lane depends on
// its neighbour, so sheared motion costs the same as uniform motion (no DPP exchange, no
// exec-masked second round, no scalar loads for lane 63).  The (u,v) pairs arrive in aligned
// register pairs: the bilinear blend runs as v_pk_mul_f32 / v_pk_fma_f32 on both components at
// once, same operation order per component as blend() - results are bit-identical to the
// one-plane-per-component path.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The field's four taps in ONE load (PAIRS): with the rows interleaved in pairs - plane
// {p(y,x), p(y+1,x)} per pixel, pack_field_rows below - the 2x2 footprint of a sample is 16
// contiguous bytes at the lane's own position, so the field costs the vector memory pipeline one
// dwordx4 instead of two dwordx2: 5 loads per pixel and lead step instead of 6 (same values, same
// blend order - bit-identical).
template <bool WITH_P, bool PAIRS>
__device__ __forceinline__ void sample_interior_packed(const Fields &F, int X, int Y, float fx, float fy,
                                                       int n, float &su, float &sv, float &sp) {
  const unsigned offp = static_cast<unsigned>(__mul24(Y, n) + X) << 2;
  const unsigned offuv = offp << 1;
  const int rb = F.row_bytes;
  const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(F.ruv, static_cast<int>(offuv), 0, 0);
  const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(F.ruv, static_cast<int>(offuv), 2 * rb, 0);
  u32x2 pt, pb;  // {p(X), p(X+1)} of the two tap rows
  if (WITH_P && PAIRS) {
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(F.rpp, static_cast<int>(offuv), 0, 0);
    pt = u32x2{q.x, q.z};  // {p(Y,X), p(Y,X+1)}
    pb = u32x2{q.y, q.w};  // {p(Y+1,X), p(Y+1,X+1)}
  } else if (WITH_P) {
    pt = __builtin_amdgcn_raw_buffer_load_b64(F.rp, static_cast<int>(offp), 0, 0);
    pb = __builtin_amdgcn_raw_buffer_load_b64(F.rp, static_cast<int>(offp), rb, 0);
  }
  const Weights w = make_weights(fx, fy);
  const f32x4 T = __builtin_bit_cast(f32x4, t), B = __builtin_bit_cast(f32x4, b);
  f32x2 acc = T.xy * w.w00;
  acc = __builtin_elementwise_fma(f32x2{w.w01, w.w01}, T.zw, acc);
  acc = __builtin_elementwise_fma(f32x2{w.w10, w.w10}, B.xy, acc);
  acc = __builtin_elementwise_fma(f32x2{w.w11, w.w11}, B.zw, acc);
  su = acc.x;
  sv = acc.y;
  if (WITH_P) {
    const f32x2 PT = __builtin_bit_cast(f32x2, pt), PB = __builtin_bit_cast(f32x2, pb);
    sp = blend(w, PT.x, PT.y, PB.x, PB.y);
  }
}

// {u,v} interleaved copy of the velocity planes: 4 pixels per thread, dwordx4 in and out
__global__ __launch_bounds__(256) void pack_velocity(const float *__restrict__ vel, float *__restrict__ uv,
                                                     size_t plane) {
  const size_t i = (static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i + 3 < plane) {
    const f32x4 u = *reinterpret_cast<const f32x4 *>(vel + i);
    const f32x4 v = *reinterpret_cast<const f32x4 *>(vel + plane + i);
    f32x4 *o = reinterpret_cast<f32x4 *>(uv + 2 * i);
    o[0] = f32x4{u.x, v.x, u.y, v.y};
    o[1] = f32x4{u.z, v.z, u.w, v.w};
  } else {
    for (size_t k = i; k < plane; ++k) {
      uv[2 * k] = vel[k];
      uv[2 * k + 1] = vel[plane + k];
    }
  }
}

// row-pair copy of the field: out[(y n + x) 2 + {0,1}] = {p(y,x), p(min(y+1, m-1), x)}; 4 pixels per thread
__global__ __launch_bounds__(256) void pack_field_rows(const float *__restrict__ p, float *__restrict__ out, int m,
                                                       int n) {
  const int y = blockIdx.y;
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x >= n) return;
  const float *r0 = p + static_cast<size_t>(y) * n, *r1 = p + static_cast<size_t>(min(y + 1, m - 1)) * n;
  float *o = out + (static_cast<size_t>(y) * n + x) * 2;
  if (x + 3 < n && (n & 3) == 0) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(r0 + x), b = *reinterpret_cast<const f32x4 *>(r1 + x);
    reinterpret_cast<f32x4 *>(o)[0] = f32x4{a.x, b.x, a.y, b.y};
    reinterpret_cast<f32x4 *>(o)[1] = f32x4{a.z, b.z, a.w, b.w};
  } else {
    for (int k = 0; k < 4 && x + k < n; ++k) {
      o[2 * k] = r0[x + k];
      o[2 * k + 1] = r1[x + k];
    }
  }
}

// ---- general path (some lane touches the border) ------------------------------
// velocity, mode="nearest": the coordinate is clamped to [0,len-1]; with clamped
// indices both taps coincide outside the range, which gives the same value.
__device__ __forceinline__ void sample_velocity_border(const Fields &F, int X, int Y, float fx,
                                                       float fy, int m, int n, float &su,
                                                       float &sv) {
  const int x0 = min(max(X, 0), n - 1), x1 = min(max(X + 1, 0), n - 1);
  const int y0 = min(max(Y, 0), m - 1), y1 = min(max(Y + 1, 0), m - 1);
  const unsigned r0 = static_cast<unsigned>(__mul24(y0, n)), r1 = static_cast<unsigned>(__mul24(y1, n));
  const unsigned o00 = (r0 + x0) << 2, o01 = (r0 + x1) << 2, o10 = (r1 + x0) << 2,
                 o11 = (r1 + x1) << 2;
  const float a = ld(F.u0, o00), b = ld(F.u0, o01), c = ld(F.u0, o10), d = ld(F.u0, o11);
  const float e = ld(F.v0, o00), f = ld(F.v0, o01), g = ld(F.v0, o10), h = ld(F.v0, o11);
  const Weights w = make_weights(fx, fy);
  su = blend(w, a, b, c, d);
  sv = blend(w, e, f, g, h);
}

template <int ORDER, bool GEN>
__device__ __forceinline__ float sample_precip_off_fast(const float *p, int X, int Y, float fx, float fy, int m,
                                                        int n, float outval, int bmode) {
  if (GEN) return sample_precip_edge<ORDER>(p, X, Y, fx, fy, m, n, outval, bmode);
  return sample_precip_border<ORDER>(p, X, Y, fx, fy, m, n, outval);
}

// interp_order 3: the "constant" rule alone, or any boundary mode (GEN)
template <bool GEN>
__device__ __forceinline__ float sample_cubic(const Fields &F, int X, int Y, float fx, float fy, int m, int n,
                                              float outval) {
  if (GEN) return sample_precip_cubic_mode(F.coef, F.p0, X, Y, fx, fy, m, n, F.minval, outval, F.bmode, F.cpad, F.sorder);
  return sample_precip_cubic(F.coef, F.p0, X, Y, fx, fy, m, n, F.minval, F.sorder);
}

// What to sample at the NPX positions of a thread
enum : int { kVel = 1, kPrecip = 2 };

template <int NPX, int ORDER, int WHAT, int MODE, bool GEN>
__device__ __forceinline__ void sample_at(const Fields &F, const int (&X)[NPX],
                                          const int (&Y)[NPX], const float (&fx)[NPX],
                                          const float (&fy)[NPX], int m, int n, float outval,
                                          float (&su)[NPX], float (&sv)[NPX], float (&sp)[NPX]) {
  constexpr bool kWithP = (WHAT & kPrecip) != 0;
  bool inside = true;
#pragma unroll
  for (int j = 0; j < NPX; ++j) inside = inside && wave_all_interior(X[j], Y[j], m, n);
  // wave-uniform branch: interior waves (almost all of them) skip every clamp
  if (inside) {
    if (MODE == kModePacked || MODE == kModePacked2) {
#pragma unroll
      for (int j = 0; j < NPX; ++j)
        sample_interior_packed<kWithP && ORDER == 1, MODE == kModePacked2>(F, X[j], Y[j], fx[j], fy[j], n, su[j], sv[j],
                                                                            sp[j]);
    } else {
      sample_interior<NPX, kWithP && ORDER == 1>(F, X, Y, fx, fy, n, su, sv, sp);
    }
    if (kWithP && ORDER == 0) {
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        const int xi = X[j] + (fx[j] >= 0.5f ? 1 : 0), yi = Y[j] + (fy[j] >= 0.5f ? 1 : 0);
        sp[j] = ld(F.p0, static_cast<unsigned>(__mul24(yi, n) + xi) << 2);
      }
    }
    if (kWithP && ORDER == 3) {
#pragma unroll
      for (int j = 0; j < NPX; ++j)
        sp[j] = sample_cubic<GEN>(F, X[j], Y[j], fx[j], fy[j], m, n, outval);
    }
    // keep the optimiser from sinking both branches into one load sequence with
    // selected 64-bit addresses (that would cost the fast path its addressing)
    asm volatile("" ::: "memory");
  } else {
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      if (WHAT & kVel) sample_velocity_border(F, X[j], Y[j], fx[j], fy[j], m, n, su[j], sv[j]);
      if (kWithP) {
        sp[j] = ORDER == 3 ? sample_cubic<GEN>(F, X[j], Y[j], fx[j], fy[j], m, n, outval)
                           : sample_precip_off_fast<(ORDER == 3 ? 1 : ORDER), GEN>(F.p0, X[j], Y[j], fx[j], fy[j],
                                                                          m, n, outval, F.bmode);
      }
    }
  }
}

// GEN: the field resampling honours F.bmode (any scipy boundary mode); otherwise the kernel only
// contains the "constant" rule and none of the folding code
template <int NPX, int ORDER, bool HAS_PRECIP, int MODE, bool GEN>
__global__ __launch_bounds__(kTileX *kDirectWaves) void semilag_fused(
    const float *__restrict__ precip, const float *__restrict__ vel, const float *__restrict__ vel_packed,
    const float *__restrict__ field_pairs, float *__restrict__ out,
    double *__restrict__ disp, const float *__restrict__ scale, float first_scale, int m, int n,
    int T, int n_iter, int resume, float outval, int row0, int rows, const float *__restrict__ coef,
    float minval, int bmode, int coef_pad, int tiles_x, int n_tiles, int tiles_per_xcd) {
  // XCD-aware remap: hardware block b -> XCD b % 8; give XCD k the k-th band of tiles
  const int b = blockIdx.x;
  const int tile = (b % kNumXcd) * tiles_per_xcd + b / kNumXcd;
  if (tile >= n_tiles) return;
  // threads past the right/bottom edge shadow the edge pixel (all 64 lanes stay
  // active for the cross-lane exchange); only their stores are masked
  const int xt = (tile % tiles_x) * kTileX + (threadIdx.x & (kTileX - 1));
  // row band [row0, row0 + rows) of the image (the whole image unless the output is tiled)
  constexpr int kWaves = kDirectWaves;
  const int yt = row0 + (tile / tiles_x) * (kWaves * NPX) + (threadIdx.x / kTileX) * NPX;
  const int x = min(xt, n - 1);
  const size_t plane = static_cast<size_t>(m) * n;
  Fields F;
  F.u0 = vel;
  F.v0 = vel + plane;
  F.p0 = precip;
  const int plane_bytes = static_cast<int>(plane * sizeof(float));
  F.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel), 0, plane_bytes, 0x00020000);
  F.rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel + plane), 0, plane_bytes, 0x00020000);
  F.rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(HAS_PRECIP ? precip : vel), 0, plane_bytes,
                                           0x00020000);
  constexpr bool kPackedVel = MODE == kModePacked || MODE == kModePacked2;
  F.ruv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(kPackedVel ? vel_packed : vel), 0, 2 * plane_bytes,
                                            0x00020000);
  F.rpp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MODE == kModePacked2 ? field_pairs : vel), 0,
                                            2 * plane_bytes, 0x00020000);
  F.row_bytes = n * static_cast<int>(sizeof(float));
  F.coef = coef;
  F.cpad = coef_pad & 0xff;  // (the B-spline's order rides in the second byte: launch_variant)
  F.sorder = (coef_pad >> 8) != 0 ? (coef_pad >> 8) : 3;
  F.minval = minval;
  F.bmode = bmode;

  // trajectory state per pixel: absolute integer position + fraction, and the increment
  int y[NPX], px[NPX], py[NPX];
  float fx[NPX], fy[NPX], vix[NPX], viy[NPX], su[NPX], sv[NPX], sp[NPX];
  bool live[NPX];
  unsigned pix[NPX], opix[NPX];
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    live[j] = xt < n && yt + j < row0 + rows;
    y[j] = min(yt + j, m - 1);
    pix[j] = static_cast<unsigned>(__mul24(y[j], n) + x) << 2;
    opix[j] = static_cast<unsigned>(__mul24(y[j] - row0, n) + x) << 2;  // output is band-local
    px[j] = x;
    py[j] = y[j];
    fx[j] = fy[j] = sp[j] = 0.f;
  }

  if (resume) {
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      const double dx = disp[static_cast<size_t>(y[j]) * n + x];
      const double dy = disp[plane + static_cast<size_t>(y[j]) * n + x];
      split_displacement(dx, px[j], fx[j]);
      split_displacement(dy, py[j], fy[j]);
    }
    // resume == 1: a displacement_prev in the reference's sense - the increment is sampled where the trajectory
    // stands (:204-207).  resume == 2: the buffer holds the BASE positions of a custom xy_coords grid relative to the
    // integer grid (:174-179, :221), no previous displacement: the increment starts as the grid's own velocity (:203)
    if (resume == 1) {
      sample_at<NPX, ORDER, kVel, MODE, GEN>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
      const float s0 = scale[0];
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        vix[j] = su[j] * s0;
        viy[j] = sv[j] * s0;
      }
    }
  }
  if (resume != 1) {
    // first increment is NOT divided by n_iter (semilagrangian.py:202)
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      vix[j] = ld(F.u0, pix[j]) * first_scale;
      viy[j] = ld(F.v0, pix[j]) * first_scale;
    }
  }

  // what a lost trajectory (NaN coordinate) samples: scipy gives cval except where it interpolates
  // across the NaN ("nearest", "grid-constant" with order >= 1; order 3 masks it to NaN anyway)
  const float lostval = (ORDER == 3 || bmode == kModeNearest || (ORDER == 1 && bmode == kModeGridConstant))
                            ? __builtin_nanf("")
                            : outval;
  // with n_iter > 0 the increment is only ever used halved (midpoint rule): carry Vi / 2,
  // which is the same number as halving at the point of use (scaling by 2 is exact)
  if (n_iter > 0) {
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      vix[j] *= 0.5f;
      viy[j] *= 0.5f;
    }
  }

  for (int t = 0; t < T; ++t) {
    const float s = scale[t];  // (lead-time increment / vel_timestep) / max(n_iter, 1)
    if (n_iter > 0) {
      const float half_s = 0.5f * s;
      for (int k = 0; k < n_iter; ++k) {
        int mx[NPX], my[NPX];
        float gx[NPX], gy[NPX];
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
          mx[j] = px[j];
          my[j] = py[j];
          gx[j] = fx[j];
          gy[j] = fy[j];
          retreat(mx[j], gx[j], vix[j]);  // midpoint rule (:213), vix = Vi / 2
          retreat(my[j], gy[j], viy[j]);
        }
        sample_at<NPX, ORDER, kVel, MODE, GEN>(F, mx, my, gx, gy, m, n, outval, su, sv, sp);
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
          retreat(px[j], fx[j], su[j] * s);
          retreat(py[j], fy[j], sv[j] * s);
        }
        if (HAS_PRECIP && k == n_iter - 1) {
          sample_at<NPX, ORDER, kVel | kPrecip, MODE, GEN>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
        } else {
          sample_at<NPX, ORDER, kVel, MODE, GEN>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
        }
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
          vix[j] = su[j] * half_s;
          viy[j] = sv[j] * half_s;
        }
      }
    } else {
      if (t > 0 || resume == 1) {  // (:216: ti > 0 or a displacement_prev; base positions alone do not count)
        sample_at<NPX, ORDER, kVel, MODE, GEN>(F, px, py, fx, fy, m, n, outval, su, sv, sp);
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
          vix[j] = su[j] * s;
          viy[j] = sv[j] * s;
        }
      }
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        retreat(px[j], fx[j], vix[j]);
        retreat(py[j], fy[j], viy[j]);
      }
      if (HAS_PRECIP) {
        bool inside = true;
#pragma unroll
        for (int j = 0; j < NPX; ++j) inside = inside && wave_all_interior(px[j], py[j], m, n);
        if (ORDER == 1 && inside) {
          float v[NPX][4];
#pragma unroll
          for (int j = 0; j < NPX; ++j) {
            const unsigned off = static_cast<unsigned>(__mul24(py[j], n) + px[j]) << 2;
            v[j][0] = ld(F.p0, off);
            v[j][1] = ld(F.p0, off, 1);
            v[j][2] = ld(F.p0, off + static_cast<unsigned>(F.row_bytes));
            v[j][3] = ld(F.p0, off + static_cast<unsigned>(F.row_bytes), 1);
          }
#pragma unroll
          for (int j = 0; j < NPX; ++j)
            sp[j] = blend(make_weights(fx[j], fy[j]), v[j][0], v[j][1], v[j][2], v[j][3]);
          asm volatile("" ::: "memory");
        } else {
#pragma unroll
          for (int j = 0; j < NPX; ++j)
            sp[j] = ORDER == 3
                        ? sample_cubic<GEN>(F, px[j], py[j], fx[j], fy[j], m, n, outval)
                        : sample_precip_off_fast<(ORDER == 3 ? 1 : ORDER), GEN>(F.p0, px[j], py[j], fx[j], fy[j], m,
                                                                       n, outval, F.bmode);
        }
      }
    }
    if (HAS_PRECIP) {
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        // Non-finite velocities (allow_nonfinite_values, semilagrangian.py:106-137): a trajectory that
        // sampled one carries a NaN fraction from then on (v_cvt_flr(NaN) = 0 keeps the integer part in
        // range, every later sample is NaN).  map_coordinates answers a NaN coordinate with cval in
        // the "constant" mode (and the folding modes), with NaN where it interpolates across it.
        sp[j] = lost(fx[j], fy[j]) ? lostval : sp[j];
        // streamed once, never re-read: keep the output out of the L2 ways the input planes live in
        if (live[j]) {
          __builtin_nontemporal_store(sp[j], reinterpret_cast<float *>(reinterpret_cast<char *>(out) + opix[j]));
        }
      }
      out += static_cast<size_t>(rows) * n;
    }
  }

  if (disp != nullptr) {
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      if (!live[j]) continue;
      disp[static_cast<size_t>(y[j]) * n + x] =
          static_cast<double>(px[j]) - static_cast<double>(x) + static_cast<double>(fx[j]);
      disp[plane + static_cast<size_t>(y[j]) * n + x] =
          static_cast<double>(py[j]) - static_cast<double>(y[j]) + static_cast<double>(fy[j]);
    }
  }
}

// ---- workgroup window ------------------------------------------------------------------------------
// The gather kernels above are bound by the CU's vector-memory pipeline: five wave64 dwordxw gathers per
// pixel and lead step return 80 B per lane through a 64 B/clk path (docs/history.md 3.1), whatever the caches
// hold.  The window kernel below takes that pipeline out of the inner loop.  A workgroup of eight waves
// owns a 64 x 32 tile (four rows per lane) and keeps a WINDOW of the motion field and of the advected
// field - 96 x 64 texels around the tile's current sample positions - in LDS ACROSS lead steps: a sampling
// pass is LDS reads and arithmetic, nothing else.  The trajectory is carried relative to the window
// (pre-scaled column offset, row offset), so the two unsigned compares that prove "all four taps inside
// the window" replace the image-interior test and the LDS address is one multiply-add.
//  * A wave whose samples are not all inside the window (border of the image, extreme deformation, a
//    lost trajectory) takes that pass through the gathers of the one-plane-per-component path - wave by
//    wave, pass by pass, same values, same blend order: results are bit-identical to every other kernel.
//  * Once per lead step the waves agree (one s_barrier) on whether the window has to move: a wave asks
//    for it when the corner samples of its patch come closer to the window's edge than the distance the
//    next step covers.  The new window is placed with its slack AHEAD of the motion (it then lasts
//    slack / speed lead steps: ~5 at 6 px per step), filled by coalesced dwordxw loads + ds_write_b128,
//    and the lanes re-base their offsets.
//  * XCD cells, guards, what bounds the kernel now and everything round 5 measured on the way: DESIGN.md 3.1 / 9.

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) int lds_int;

// LDS traffic only: the nontemporal output stores of the lead step stay in flight across the barrier
__device__ __forceinline__ void win_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ int smin(int a, int b) {
  int r;
  asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
  return r;
}
__device__ __forceinline__ int smax(int a, int b) {
  int r;
  asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc");
  return r;
}

// debug counters of the window kernel (PYSTEPS_HIP_SL_STATS=1): printed after every launch, which then waits
static unsigned long long *g_win_stats = nullptr;

// ---- the window's geometry and control block ----------------------------------------------------------
template <int WAVES, int WW, int WH, int ROWS = 4, int OCC = 4>
struct WinCfg {
  static constexpr int kWaves = WAVES, kW = WW, kH = WH;
  static constexpr int kRows = ROWS;  // image rows per lane
  static constexpr int kOcc = OCC;    // waves per SIMD the register budget is cut for (two workgroups of 8 waves: 4)
  // u and v are interleaved as {u,v} pairs in ONE window plane (8-byte texels)
  static constexpr int kTileY = ROWS * WAVES;
  static constexpr unsigned kPitch4 = WW * 4u;          // bytes per window row of one plane
  static constexpr unsigned kPlaneBytes = WW * WH * 4u;
  static constexpr int kItems = WH * (WW / 4);          // 16-byte items per plane
  static constexpr int kThreads = kTileX * WAVES;
  static constexpr unsigned kCtlVel = 16u * WAVES, kCtlFlag = kCtlVel + 8u;  // byte offsets in the control block
  static constexpr unsigned kCtlTile = kCtlFlag + 12u;  // the tile a persistent workgroup works on
  static constexpr int kCtlWords = 4 * WAVES + 2 + 3 + 3;
};
using Win8 = WinCfg<8, 96, 64>;
// (Round 6 measured two rows per lane at six waves per SIMD on the same box: twelve waves on a 64 x 24 tile with a
// 96 x 56 window, two workgroups per CU - WinCfg<12, 96, 56, 2, 6> - 1.29 / 1.23 ms (sheared / uniform field), eight
// waves on a 64 x 16 tile with a 96 x 42 window, three per CU - WinCfg<8, 96, 42, 2, 6> - 1.21 / 1.13, against 1.145 /
// 1.077 for this one: more waves do not buy what the smaller tiles' extra fills, barriers and per-wave scalar work
// cost - the kernel is not short of latency hiding.  profiles/r06/a_window_order_persist_ab.txt)

struct Window {
  unsigned uv, p;    // LDS byte addresses of the {u,v} plane and of the field plane
  unsigned ctl;
  int ox, oy;        // image position of the window's first texel (uniform over the workgroup)
  // where the corner samples of a patch may be without asking for a new window (pre-scaled columns / rows):
  // set when a window is placed, from the direction and speed of travel
  int lo_x, hi_x, lo_y, hi_y;
  // the window reaches beyond the image (uniform): its texels out there hold what the border rules read - the motion
  // field's edge values replicated, the advected field's mirror texel at index len - and a field sample may lie outside
  int edge;
  // rows of the planes start on 16-byte boundaries (n % 4 == 0, 16-byte aligned planes): the fills move dwordx4 items
  int fill16;
  unsigned long long *stats;
};

__device__ __forceinline__ void win_count(const Window &W, int which) {
  if (W.stats != nullptr && (threadIdx.x & 63) == 0) atomicAdd(W.stats + which, 1ull);
}

// the box of a patch's corner samples (window-relative, pre-scaled columns / rows): eight v_readlane, scalar min / max
struct WinBox {
  int lox, hix, loy, hiy;
};
template <int kWinRows>
__device__ __forceinline__ WinBox win_corners(const int (&dxw)[kWinRows], const int (&dy)[kWinRows]) {
  const int xa = __builtin_amdgcn_readlane(dxw[0], 0), xb = __builtin_amdgcn_readlane(dxw[0], 63);
  const int xc = __builtin_amdgcn_readlane(dxw[kWinRows - 1], 0), xd = __builtin_amdgcn_readlane(dxw[kWinRows - 1], 63);
  const int ya = __builtin_amdgcn_readlane(dy[0], 0), yb = __builtin_amdgcn_readlane(dy[0], 63);
  const int yc = __builtin_amdgcn_readlane(dy[kWinRows - 1], 0), yd = __builtin_amdgcn_readlane(dy[kWinRows - 1], 63);
  return {smin(smin(xa, xb), smin(xc, xd)), smax(smax(xa, xb), smax(xc, xd)), smin(smin(ya, yb), smin(yc, yd)),
          smax(smax(ya, yb), smax(yc, yd))};
}

// lane 0: the wave's box (one 16-byte store), wave 0 also its direction of travel
template <class C>
__device__ __forceinline__ void win_publish(const Window &W, int wave, const WinBox &bx, float vx_lane, float vy_lane) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) i32x4 lds_i32x4;
  lds_int *ctl = (lds_int *)(size_t)W.ctl;
  *(lds_i32x4 *)(size_t)(W.ctl + 16u * wave) = i32x4{bx.lox, bx.hix, bx.loy, bx.hiy};
  if (wave == 0) {
    ctl[C::kCtlVel / 4 + 0] = __float_as_int(vx_lane);
    ctl[C::kCtlVel / 4 + 1] = __float_as_int(vy_lane);
  }
}

// Every wave of the workgroup, with every wave's box published and nobody reading the window any more: place the
// new window ahead of the motion, re-base the offsets, fill it.  (The caller orders the LDS writes before the next reader.)
template <class C, bool GEN>
__device__ __forceinline__ void win_place_and_fill(const Fields &F, Window &W, bool force, int (&dxw)[C::kRows],
                                                    int (&dy)[C::kRows], float move_scale, int m, int n) {
  constexpr int kWinRows = C::kRows;
  lds_int *ctl = (lds_int *)(size_t)W.ctl;
  int ulox = 0x7fffffff, uhix = -0x7fffffff, uloy = 0x7fffffff, uhiy = -0x7fffffff;
#pragma unroll
  for (int w = 0; w < C::kWaves; ++w) {
    ulox = min(ulox, ctl[w * 4 + 0]);
    uhix = max(uhix, ctl[w * 4 + 1]);
    uloy = min(uloy, ctl[w * 4 + 2]);
    uhiy = max(uhiy, ctl[w * 4 + 3]);
  }
  const float wvx = __int_as_float(ctl[C::kCtlVel / 4 + 0]), wvy = __int_as_float(ctl[C::kCtlVel / 4 + 1]);
  // first and last texel the tile touches now (right / lower tap included), relative to the current origin
  const int bx0 = ulox, bx1 = uhix + 1, by0 = uloy, by1 = uhiy + 1;
  const int slack_x = max(C::kW - (bx1 - bx0 + 1), 0), slack_y = max(C::kH - (by1 - by0 + 1), 0);
  // texels kept on the low side: all the slack but two where the motion goes that way (a positive velocity moves
  // the samples towards lower coordinates), two where it comes from, half of it in calm air
  const int keep_x = wvx > 0.125f ? max(slack_x - 2, 0) : (wvx < -0.125f ? min(slack_x, 2) : slack_x / 2);
  const int keep_y = wvy > 0.125f ? max(slack_y - 2, 0) : (wvy < -0.125f ? min(slack_y, 2) : slack_y / 2);
  // 16-byte aligned rows.  With the "constant" rule for the advected field (every instantiation but GEN) the window
  // FOLLOWS the samples out of the image: out there its texels are filled with what the border rules read (below), so
  // a tile whose trajectories leave the image keeps sampling from LDS instead of taking every pass through the gathers
  // (round 6; 6 % of the kernel's time at 4096^2).  The other boundary modes keep the window inside the image: their
  // samples out there go through sample_at<>.
  constexpr int kFar = 1 << 27;  // (a saturated box - a trajectory at the end of the number line - places the window nowhere useful)
  const int want_x = min(max(sat_add(W.ox, sat_sub(bx0, keep_x)), -kFar), kFar) & ~3;
  const int want_y = min(max(sat_add(W.oy, sat_sub(by0, keep_y)), -kFar), kFar);
  const int nox = rfl(GEN ? min(max(want_x, 0), n - C::kW) : want_x);  // (n - kW is a multiple of 4 where n is)
  const int noy = rfl(GEN ? min(max(want_y, 0), m - C::kH) : want_y);
  const bool edge = !GEN && (nox < 0 || nox + C::kW > n || noy < 0 || noy + C::kH > m);  // (uniform)
  PSH_DASSERT((W.fill16 == 0 || (nox & 3) == 0) && (edge || (nox >= 0 && nox + C::kW <= n && noy >= 0 && noy + C::kH <= m)));  // the window lies in the image
  // the room a patch needs ahead of its corner samples before the next lead step (the distance the last one covered + 2);
  // a lost trajectory (NaN) asks for nothing
  const float mvx = fabsf(wvx) < 64.f ? fabsf(wvx) * move_scale + 2.f : 2.f, mvy = fabsf(wvy) < 64.f ? fabsf(wvy) * move_scale + 2.f : 2.f;
  const int gx = rfl(static_cast<int>(mvx)), gy = rfl(static_cast<int>(mvy));
  W.lo_x = wvx > 0.f ? gx : 1;
  W.hi_x = C::kW - 2 - (wvx > 0.f ? 1 : gx);
  W.lo_y = wvy > 0.f ? gy : 1;
  W.hi_y = C::kH - 2 - (wvy > 0.f ? 1 : gy);
  // (a tile parked at the image border keeps asking: the window it would get is the one it has)
  if (!force && nox == W.ox && noy == W.oy) return;
  win_count(W, 2);
  const int ddx = nox - W.ox, ddy = noy - W.oy;
#pragma unroll
  for (int j = 0; j < kWinRows; ++j) {
    dxw[j] = sat_sub(dxw[j], ddx);  // (a trajectory that left for good stays at the end of the number line)
    dy[j] = sat_sub(dy[j], ddy);
  }
  W.ox = nox;
  W.oy = noy;
  W.edge = edge ? 1 : 0;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // opaque: the item addresses below are not loop invariants worth their registers
  constexpr int kRounds = (C::kItems + C::kThreads - 1) / C::kThreads;
  // (threads past the last item repeat it: the same bytes to the same place, and no exec-masked rounds)
  u32x4 bu[kRounds], bv[kRounds], bp[kRounds];
  if (!edge) {
    // (rows that do not start on 16-byte boundaries - n % 4 != 0 - load their items from dword-aligned addresses: a
    // dwordx4 needs no more; round 6, such shapes took the gather kernels before)
    const unsigned org = static_cast<unsigned>(noy) * static_cast<unsigned>(n) + static_cast<unsigned>(nox);
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int item = min(tid + C::kThreads * k, C::kItems - 1);
      const int row = item / (C::kW / 4), c = item - row * (C::kW / 4);
      const int off = static_cast<int>((org + row * n + 4 * c) << 2);
      bu[k] = __builtin_amdgcn_raw_buffer_load_b128(F.ru, off, 0, 0);
      bv[k] = __builtin_amdgcn_raw_buffer_load_b128(F.rv, off, 0, 0);
      bp[k] = __builtin_amdgcn_raw_buffer_load_b128(F.rp, off, 0, 0);
    }
  } else if (!W.fill16) {
    // A window that reaches beyond an image whose rows do not start on 16-byte boundaries (n % 4 != 0: a 640 x 710
    // composite, say): texel by texel, every one with its own clamped / mirrored column - the rules of the branch below,
    // which here also cover an item that straddles the image's edge.  Twelve dword loads per item instead of three
    // dwordx4, in border tiles only.
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int item = min(tid + C::kThreads * k, C::kItems - 1);
      const int row = item / (C::kW / 4), c = item - row * (C::kW / 4);
      const int r = noy + row, x0 = nox + 4 * c;
      const int rv = min(max(r, 0), m - 1), rq = r == m ? m - 2 : rv;
      const unsigned base_v = static_cast<unsigned>(__mul24(rv, n)), base_q = static_cast<unsigned>(__mul24(rq, n));
      unsigned tu[4], tv[4], tq[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = x0 + e;
        const int xv = min(max(x, 0), n - 1), xq = x == n ? n - 2 : xv;
        tu[e] = __builtin_amdgcn_raw_buffer_load_b32(F.ru, static_cast<int>((base_v + xv) << 2), 0, 0);
        tv[e] = __builtin_amdgcn_raw_buffer_load_b32(F.rv, static_cast<int>((base_v + xv) << 2), 0, 0);
        tq[e] = __builtin_amdgcn_raw_buffer_load_b32(F.rp, static_cast<int>((base_q + xq) << 2), 0, 0);
      }
      bu[k] = u32x4{tu[0], tu[1], tu[2], tu[3]};
      bv[k] = u32x4{tv[0], tv[1], tv[2], tv[3]};
      bp[k] = u32x4{tq[0], tq[1], tq[2], tq[3]};
    }
  } else {
    // A window that reaches beyond the image.  Motion field, mode "nearest" (sample_velocity_border): every tap index is
    // clamped on its own - texel (r, x) holds the value at (clamp r, clamp x).  Advected field, mode "constant"
    // (sample_precip_border): a sample outside [0, len - 1] is outval whatever the taps are (win_sample decides that
    // from the position), and the one in-range sample that touches index len - coordinate len - 1 exactly, weight 0 -
    // reads the MIRROR texel len - 2 (0 x NaN matters): texel len holds the value at len - 2.  n % 4 == 0 and the
    // window's columns start at a multiple of 4, so a 16-byte item lies inside the image's columns or outside as a whole.
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
      const int item = min(tid + C::kThreads * k, C::kItems - 1);
      const int row = item / (C::kW / 4), c = item - row * (C::kW / 4);
      const int r = noy + row, x = nox + 4 * c;
      const int rv = min(max(r, 0), m - 1), rq = r == m ? m - 2 : rv;
      const int xi = min(max(x, 0), n - 4);
      const int off_v = static_cast<int>(static_cast<unsigned>(__mul24(rv, n) + xi) << 2);
      const int off_q = static_cast<int>(static_cast<unsigned>(__mul24(rq, n) + xi) << 2);
      u32x4 tu = __builtin_amdgcn_raw_buffer_load_b128(F.ru, off_v, 0, 0);
      u32x4 tv = __builtin_amdgcn_raw_buffer_load_b128(F.rv, off_v, 0, 0);
      u32x4 tq = __builtin_amdgcn_raw_buffer_load_b128(F.rp, off_q, 0, 0);
      if (x < 0) {  // left of the image: column 0
        tu = u32x4{tu.x, tu.x, tu.x, tu.x};
        tv = u32x4{tv.x, tv.x, tv.x, tv.x};
      } else if (x >= n) {  // right of it: column n - 1 for the motion field, n - 2 (the mirror of n) for the advected field
        tu = u32x4{tu.w, tu.w, tu.w, tu.w};
        tv = u32x4{tv.w, tv.w, tv.w, tv.w};
        tq = u32x4{tq.z, tq.z, tq.z, tq.z};
      }
      bu[k] = tu;
      bv[k] = tv;
      bp[k] = tq;
    }
  }
#pragma unroll
  for (int k = 0; k < kRounds; ++k) {
    const unsigned l = 16u * min(tid + C::kThreads * k, C::kItems - 1);
    // four pixels of the {u,v} plane = two 16-byte items
    *(lds_u32x4 *)(size_t)(W.uv + 2u * l) = u32x4{bu[k].x, bv[k].x, bu[k].y, bv[k].y};
    *(lds_u32x4 *)(size_t)(W.uv + 2u * l + 16u) = u32x4{bu[k].z, bv[k].z, bu[k].w, bv[k].w};
    *(lds_u32x4 *)(size_t)(W.p + l) = bp[k];
  }
}

// Once per lead step, every wave of the workgroup: publish the box of the patch's corner samples, ask for a new
// window if they are about to leave this one, agree at ONE barrier, and if anybody asked: place, re-base, fill,
// second barrier.  `phase` cycles through three flag words so that clearing the next one never races with a wave
// that still has to read it.
template <class C, bool GEN>
__device__ __forceinline__ void win_update(const Fields &F, Window &W, int phase, bool force, int (&dxw)[C::kRows],
                                            int (&dy)[C::kRows], float vx_lane, float vy_lane, float move_scale, int m,
                                            int n) {
  const int lane = threadIdx.x & 63, wave = rfl(static_cast<int>(threadIdx.x >> 6));
  lds_int *ctl = (lds_int *)(size_t)W.ctl;
  const WinBox bx = win_corners(dxw, dy);
  const bool near = force || bx.lox < W.lo_x || bx.hix > W.hi_x || bx.loy < W.lo_y || bx.hiy > W.hi_y;
  if (lane == 0) {
    win_publish<C>(W, wave, bx, vx_lane, vy_lane);
    if (near) ctl[C::kCtlFlag / 4 + phase] = 1;
    if (wave == 0) ctl[C::kCtlFlag / 4 + (phase == 2 ? 0 : phase + 1)] = 0;
  }
  win_barrier();
  if (rfl(ctl[C::kCtlFlag / 4 + phase]) == 0) return;
  // every wave is past the barrier: nobody reads the old window any more.  The boxes of this step are overwritten
  // after the barrier below - every wave has read them by then
  win_place_and_fill<C, GEN>(F, W, force, dxw, dy, move_scale, m, n);
  win_barrier();
}

// (Round 5 also built the same agreement WITHOUT the per-lead-step barrier - waves running free, a wave that needs a
// new window raising an epoch flag and waiting at an LDS-counter rendezvous the others join when they finish their own
// lead step: bit-identical, but 1.220 against 1.172 ms - waves that drift a step or two apart spread their samples over
// a larger box, the window's slack shrinks and it is refilled twice as often (9.6 instead of 5 fills per workgroup;
// profiles/r05/f_window_free_running_timings.txt).  The barrier is what keeps the window economical.)

// ---- sampling from the window -------------------------------------------------------------------------
// The velocity taps are ds_read_b64 {u,v} pairs, the field taps ds_read_b32 - single-address reads (2 LDS cycles per
// wave-instruction; the two-address forms the compiler would merge them into take 8 / 4), invisible to the compiler's
// counters: one s_waitcnt with every destination tied to it.  The trajectory update and the weights run as
// v_pk_*_f32 on the (x, y) pair of a pixel, the velocity blend on its (u, v) pair with the weights broadcast by
// op_sel, same operations, same order, same rounding per component (no contraction in this file) - bit-identical
// with the gather kernels - at 13 LDS reads and 58 VALU instructions per pixel and lead step (round 5: 63).
// Round 6: everything that does not depend on the taps runs between the issue of the reads and the wait
// (the weights; as register pairs over two pixels in the pass that also reads the field), the first pixel pair is
// blended once ITS taps are back, the output goes through a buffer descriptor.  Build-time knobs for same-box A/B runs
// (tools/build_variant.sh ... -D<knob>): PSH_WIN_NO_PAIR_WEIGHTS, PSH_WIN_NO_STAGED_WAIT, PSH_WIN_MASKED_STORES switch
// the three steps off again (profiles/r06/q_window_pairs_staged_buffer_ab.txt); all forms are bit-identical.
// (Round 5 measured two other forms of the same window: u, v and the field as three planes with every
// floating-point operation packed over the two vertically adjacent pixels of a lane - 21 LDS reads, 54 VALU, but
// three times the LDS-issue stalls: 1.200 against 1.158 ms - and the gather kernel's arithmetic on a {u,v} window - 73 VALU: 1.28 ms;
// profiles/r05/g_window_hybrid_timings.txt, b_*, c_*.)
template <class C, int WHAT, bool GEN>
__device__ __forceinline__ void win_sample(const Fields &F, const Window &W, const int (&dxw)[C::kRows],
                                            const int (&dy)[C::kRows], const f32x2 (&f)[C::kRows], int m, int n,
                                            float outval, f32x2 (&s_uv)[C::kRows], float (&sp)[C::kRows]) {
  constexpr int kWinRows = C::kRows;
  static_assert(kWinRows == 2 || kWinRows == 4, "rows per lane");
  constexpr bool kWithP = (WHAT & kPrecip) != 0;
  unsigned mx = static_cast<unsigned>(dxw[0]), my = static_cast<unsigned>(dy[0]);
#pragma unroll
  for (int j = 1; j < kWinRows; ++j) {
    mx = max(mx, static_cast<unsigned>(dxw[j]));
    my = max(my, static_cast<unsigned>(dy[j]));
  }
  // (two ballots and a scalar AND: the compiler turns the ballot of `a && b` into a select and a second compare)
  const unsigned long long okx = __builtin_amdgcn_ballot_w64(mx <= static_cast<unsigned>(C::kW - 2));
  const unsigned long long oky = __builtin_amdgcn_ballot_w64(my <= static_cast<unsigned>(C::kH - 2));
  if ((okx & oky) == __builtin_amdgcn_ballot_w64(true)) {
    win_count(W, 0);
    f32x2 t[kWinRows][4];
    float rp[kWinRows][4];
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      // texel index in the window (one v_mad_u32_u24), scaled to the 8-byte {u,v} texels and the 4-byte field texels
      const unsigned tx = __umul24(static_cast<unsigned>(dy[j]), static_cast<unsigned>(C::kW)) + static_cast<unsigned>(dxw[j]);
      // fast path only: all four taps of the pixel inside the window's planes
      PSH_DASSERT(static_cast<unsigned>(dxw[j]) <= static_cast<unsigned>(C::kW - 2) && static_cast<unsigned>(dy[j]) <= static_cast<unsigned>(C::kH - 2) &&
                  tx + C::kW + 1 < static_cast<unsigned>(C::kW * C::kH));
      const unsigned a = (tx << 3) + W.uv;
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:%c5\n\t"
                   "ds_read_b64 %3, %4 offset:%c6"
                   : "=&v"(t[j][0]), "=&v"(t[j][1]), "=&v"(t[j][2]), "=&v"(t[j][3])
                   : "v"(a), "i"(2u * C::kPitch4), "i"(2u * C::kPitch4 + 8u));
      if (kWithP) {
        const unsigned ap = (tx << 2) + W.p;
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:%c5\n\t"
                     "ds_read_b32 %3, %4 offset:%c6"
                     : "=&v"(rp[j][0]), "=&v"(rp[j][1]), "=&v"(rp[j][2]), "=&v"(rp[j][3])
                     : "v"(ap), "i"(C::kPitch4), "i"(C::kPitch4 + 4u));
      }
    }
    // make_weights(): gx = 1 - fx, gy = 1 - fy; {gy gx, gy fx, fy gx, fy fx} (products commute bit for bit).  They only
    // need the fractions: computed while the reads are in flight (round 6: 0.98 -> 0.945 ms in the bench's leg,
    // profiles/r06/n_window_early_weights_ab.txt).  The two scalar products as asm: left to the compiler the eight of a
    // lane's four pixels become packed multiplies fed by a dozen register moves; an empty volatile asm that reads the
    // pair keeps everything in front of the wait.  (Not in the instantiation for the other boundary modes, which has no
    // registers to spare during the wait: two would spill.)
    f32x2 w_mid[kWinRows];  // (fx gy, fy gx) = (w01, w10)
    float w_00[kWinRows], w_11[kWinRows];
#ifndef PSH_WIN_NO_PAIR_WEIGHTS
    // the sample that also reads the field: every weight as a register PAIR over two vertically adjacent pixels - what
    // the packed blend of the field's taps multiplies with (the compiler packs that blend itself and otherwise builds
    // these pairs by eight register moves BEHIND the wait); the velocity blends pick their half by op_sel
    constexpr bool kPairs = !GEN && kWithP && kWinRows == 4;
    f32x2 q00[2], q01[2], q10[2], q11[2];
    if constexpr (kPairs) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const f32x2 fa = f[2 * pr], fb = f[2 * pr + 1];
        const f32x2 ga = 1.f - fa, gb = 1.f - fb;
        float a00, b00, a01, b01, a10, b10, a11, b11;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a00) : "v"(ga.x), "v"(ga.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b00) : "v"(gb.x), "v"(gb.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a01) : "v"(fa.x), "v"(ga.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b01) : "v"(fb.x), "v"(gb.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a10) : "v"(fa.y), "v"(ga.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b10) : "v"(fb.y), "v"(gb.x));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a11) : "v"(fa.x), "v"(fa.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b11) : "v"(fb.x), "v"(fb.y));
        q00[pr] = f32x2{a00, b00};
        q01[pr] = f32x2{a01, b01};
        q10[pr] = f32x2{a10, b10};
        q11[pr] = f32x2{a11, b11};
        // (in-out: behind this the pairs are opaque two-element values, so the velocity blends of the second pixel take
        // their half by op_sel instead of a register move)
        asm volatile("" : "+v"(q00[pr]), "+v"(q01[pr]), "+v"(q10[pr]), "+v"(q11[pr]));
      }
    } else
#else
    constexpr bool kPairs = false;
#endif
    if constexpr (!GEN) {
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        const f32x2 g = 1.f - f[j];
        w_mid[j] = f[j] * g.yx;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w_00[j]) : "v"(g.x), "v"(g.y));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w_11[j]) : "v"(f[j].x), "v"(f[j].y));
        asm volatile("" ::"v"(w_mid[j]));
      }
    }
#define PSH_TIE4(A, J) "+v"(A[J][0]), "+v"(A[J][1]), "+v"(A[J][2]), "+v"(A[J][3])
#ifndef PSH_WIN_NO_STAGED_WAIT
    if constexpr (kWinRows == 4 && !GEN) {
      // LDS answers in order: the first pixel pair's taps are there once at most the second pair's reads are
      // outstanding (the counter has four bits: of 32 reads the first 17 are the least that can be waited for)
      if (kWithP) {
        asm volatile("s_waitcnt lgkmcnt(15)" : PSH_TIE4(t, 0), PSH_TIE4(t, 1));
        asm volatile("" : PSH_TIE4(rp, 0), PSH_TIE4(rp, 1));
      } else {
        asm volatile("s_waitcnt lgkmcnt(8)" : PSH_TIE4(t, 0), PSH_TIE4(t, 1));
      }
      __builtin_amdgcn_sched_barrier(0);
    } else
#endif
    if constexpr (kWinRows == 4) {
      asm volatile("s_waitcnt lgkmcnt(0)" : PSH_TIE4(t, 0), PSH_TIE4(t, 1), PSH_TIE4(t, 2), PSH_TIE4(t, 3));
      if (kWithP) asm volatile("" : PSH_TIE4(rp, 0), PSH_TIE4(rp, 1), PSH_TIE4(rp, 2), PSH_TIE4(rp, 3));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : PSH_TIE4(t, 0), PSH_TIE4(t, 1));
      if (kWithP) asm volatile("" : PSH_TIE4(rp, 0), PSH_TIE4(rp, 1));
    }
#ifndef PSH_WIN_NO_PAIR_WEIGHTS
    if constexpr (kPairs) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int a = 2 * pr, b = 2 * pr + 1;
#ifndef PSH_WIN_NO_STAGED_WAIT
        if (pr == 1) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" : PSH_TIE4(t, 2), PSH_TIE4(t, 3));
          asm volatile("" : PSH_TIE4(rp, 2), PSH_TIE4(rp, 3));
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
        f32x2 acc = t[a][0] * f32x2{q00[pr].x, q00[pr].x};  // the order of sample_interior_packed
        acc = __builtin_elementwise_fma(f32x2{q01[pr].x, q01[pr].x}, t[a][1], acc);
        acc = __builtin_elementwise_fma(f32x2{q10[pr].x, q10[pr].x}, t[a][2], acc);
        acc = __builtin_elementwise_fma(f32x2{q11[pr].x, q11[pr].x}, t[a][3], acc);
        s_uv[a] = acc;
        acc = t[b][0] * f32x2{q00[pr].y, q00[pr].y};
        acc = __builtin_elementwise_fma(f32x2{q01[pr].y, q01[pr].y}, t[b][1], acc);
        acc = __builtin_elementwise_fma(f32x2{q10[pr].y, q10[pr].y}, t[b][2], acc);
        acc = __builtin_elementwise_fma(f32x2{q11[pr].y, q11[pr].y}, t[b][3], acc);
        s_uv[b] = acc;
        f32x2 ps = q00[pr] * f32x2{rp[a][0], rp[b][0]};
        ps = __builtin_elementwise_fma(q01[pr], f32x2{rp[a][1], rp[b][1]}, ps);
        ps = __builtin_elementwise_fma(q10[pr], f32x2{rp[a][2], rp[b][2]}, ps);
        ps = __builtin_elementwise_fma(q11[pr], f32x2{rp[a][3], rp[b][3]}, ps);
        sp[a] = ps.x;
        sp[b] = ps.y;
      }
    } else
#endif
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
#ifndef PSH_WIN_NO_STAGED_WAIT
      if constexpr (kWinRows == 4 && !GEN) {
        if (j == 2) {
          __builtin_amdgcn_sched_barrier(0);  // the first pair's blends stay in front of the second wait
          asm volatile("s_waitcnt lgkmcnt(0)" : PSH_TIE4(t, 2), PSH_TIE4(t, 3));
          if (kWithP) asm volatile("" : PSH_TIE4(rp, 2), PSH_TIE4(rp, 3));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#endif
      f32x2 w00, wmid, w11;
      if constexpr (GEN) {
        const f32x2 g = 1.f - f[j];
        w00 = g * g.yx;        // (gx gy, gy gx): w00 in both halves
        wmid = f[j] * g.yx;    // (fx gy, fy gx) = (w01, w10)
        w11 = f[j] * f[j].yx;  // (fx fy, fy fx): w11 in both halves
      } else {
        w00 = f32x2{w_00[j], w_00[j]};
        wmid = w_mid[j];
        w11 = f32x2{w_11[j], w_11[j]};
      }
      f32x2 acc = t[j][0] * f32x2{w00.x, w00.x};  // the order of sample_interior_packed
      acc = __builtin_elementwise_fma(f32x2{wmid.x, wmid.x}, t[j][1], acc);
      acc = __builtin_elementwise_fma(f32x2{wmid.y, wmid.y}, t[j][2], acc);
      acc = __builtin_elementwise_fma(f32x2{w11.x, w11.x}, t[j][3], acc);
      s_uv[j] = acc;
      if (kWithP) sp[j] = fmaf(w11.x, rp[j][3], fmaf(wmid.y, rp[j][2], fmaf(wmid.x, rp[j][1], w00.x * rp[j][0])));
    }
#undef PSH_TIE4
    if constexpr (kWithP && !GEN) {
      if (W.edge) {  // (uniform) sample_precip_border's rule: outside [0, len - 1] the sample is outval
#pragma unroll
        for (int j = 0; j < kWinRows; ++j) {
          const int X = sat_add(W.ox, dxw[j]), Y = sat_add(W.oy, dy[j]);
          const bool outside = X < 0 || Y < 0 || X > n - 1 || Y > m - 1 || (X == n - 1 && f[j].x > 0.f) || (Y == m - 1 && f[j].y > 0.f);
          sp[j] = outside ? outval : sp[j];
        }
      }
    }
  } else {
    win_count(W, 1);
    int X[kWinRows], Y[kWinRows];
    float sfx[kWinRows], sfy[kWinRows], ssu[kWinRows], ssv[kWinRows], ssp[kWinRows];
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      X[j] = sat_add(W.ox, dxw[j]);
      Y[j] = sat_add(W.oy, dy[j]);
      sfx[j] = f[j].x;
      sfy[j] = f[j].y;
      ssp[j] = 0.f;
    }
    sample_at<kWinRows, 1, WHAT, kModeDirect, GEN>(F, X, Y, sfx, sfy, m, n, outval, ssu, ssv, ssp);
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      s_uv[j] = f32x2{ssu[j], ssv[j]};
      if (kWithP) sp[j] = ssp[j];
    }
  }
}

// one pixel's trajectory, both axes in one packed subtraction: P -= floor stuff exactly as retreat() - the integer
// parts by SATURATING adds (sat_add, semilag_device.h): a garbage or sentinel velocity such as 1e20, which v_cvt_flr
// turns into INT_MAX / INT_MIN, sends the trajectory to the end of the number line, where it stays - as the reference's
// float64 position does - instead of wrapping around to the other side of the image or back into it
// (tests/test_semilag_gpu.py::test_window_kernel_on_sentinel_velocities).  (Round 5 carried the column pre-scaled by
// the texel size - one multiply-add less per LDS address - which wraps at 2^28 pixels, inside the window again.)
__device__ __forceinline__ void retreat_xy(int &PX, int &PY, f32x2 &f, f32x2 w) {
  const f32x2 t = f - w;
  int kx, ky;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(kx) : "v"(t.x));
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ky) : "v"(t.y));
  PX = sat_add(PX, kx);
  PY = sat_add(PY, ky);
  f = f32x2{__builtin_amdgcn_fractf(t.x), __builtin_amdgcn_fractf(t.y)};
}

// ---- which tile a workgroup takes ----------------------------------------------------------------------
// `order` lists the tiles in the sequence they are to be STARTED (slot s belongs to XCD s % 8: block b of a launch runs
// on XCD b % 8; -1 pads the XCDs' lists to one length).  Tiles are not equally expensive: the ones near the image
// border take their sampling passes through the gather path once their trajectories leave the image (three times
// the cost of a window pass) - and in image order the last ones to start would be the bottom rows, on their own
// at the end of the launch while the rest of the chip is idle.  The host builds the table (launch_window): rings
// of 32 pixels from the border inwards first, the interior in XCD cells (one cell per row band and column band for
// every XCD, each still a compact block for its L2).
//  * PERSIST = false (the default): one tile per workgroup, tile = order[blockIdx.x]; the hardware dispatcher starts the
//    next workgroup of an XCD wherever a slot frees up.
//  * PERSIST = true (PYSTEPS_HIP_SL_PERSIST=1, kept for measurements): two workgroups per CU stay, each pulling slots
//    from the list of its own XCD with one atomic per tile and, when that is exhausted, from the other XCDs' lists.
//    queue[0..7] are the lists' cursors, queue[8] counts the workgroups that are done; the last one out clears all nine
//    for the next launch on the stream.  Measured on one box against the same table without it (profiles/r06/
//    a_window_order_persist_ab.txt): 1.160 against 1.145 ms (sheared field), 1.09 against 1.077 (uniform) - the
//    dispatcher already balances workgroups of unequal length inside an XCD, the XCDs' lists are equal by
//    construction (cells), and the queue adds a barrier and an atomic round trip per tile.  What pays is the ORDER:
//    image order 1.185 / 1.117 ms, border rings first 1.145 / 1.077.
__device__ __attribute__((noinline)) int win_next_slot(unsigned *queue, int home, int tiles_per_xcd, const int *order) {
  for (int i = 0; i < kNumXcd; ++i) {
    const int k = (home + i) & (kNumXcd - 1);
    if (__hip_atomic_load(queue + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= static_cast<unsigned>(tiles_per_xcd)) continue;
    const unsigned l = __hip_atomic_fetch_add(queue + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (l < static_cast<unsigned>(tiles_per_xcd)) {
      const int t = order[l * kNumXcd + k];
      if (t >= 0) return t;
    }
  }
  return -1;
}

template <class C, bool GEN, bool PERSIST>
__global__ __launch_bounds__(C::kThreads, C::kOcc) void semilag_window(
    const float *__restrict__ precip, const float *__restrict__ vel, float *__restrict__ out_base, double *__restrict__ disp,
    const float *__restrict__ scale, float first_scale, int m, int n, int T, int n_iter, int resume, float outval,
    int row0, int rows, int bmode, int tiles_x, int n_tiles, int tiles_per_xcd, float guard,
    const int *__restrict__ order, unsigned *__restrict__ queue, unsigned long long *__restrict__ stats) {
  constexpr int kWinRows = C::kRows;
  const size_t plane = static_cast<size_t>(m) * n;
  Fields F;
  F.u0 = vel;
  F.v0 = vel + plane;
  F.p0 = precip;
  const int plane_bytes = static_cast<int>(plane * sizeof(float));
  F.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel), 0, plane_bytes, 0x00020000);
  F.rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel + plane), 0, plane_bytes, 0x00020000);
  F.rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(precip), 0, plane_bytes, 0x00020000);
  F.ruv = F.ru;
  F.rpp = F.ru;
  F.row_bytes = n * static_cast<int>(sizeof(float));
  F.coef = nullptr;
  F.cpad = 0;
  F.sorder = 3;
  F.minval = 0.f;
  F.bmode = bmode;

  __shared__ __attribute__((aligned(16))) float win_planes[3 * C::kW * C::kH];
  __shared__ __attribute__((aligned(16))) int win_ctl[C::kCtlWords];
  const int lane = threadIdx.x & (kTileX - 1);
  const float move_scale = guard * static_cast<float>(n_iter);
  const float lostval = (bmode == kModeNearest || bmode == kModeGridConstant) ? __builtin_nanf("") : outval;
  const int home = static_cast<int>(blockIdx.x) & (kNumXcd - 1);
  int tile = PERSIST ? -1 : order[blockIdx.x];
  int next = -1;
  if (PERSIST && threadIdx.x == 0) next = win_next_slot(queue, home, tiles_per_xcd, order);

  for (;;) {
    if (PERSIST) {
      // thread 0 hands the tile it pulled while the last one was running to the workgroup; nobody is inside the last
      // tile's window any more once every wave is here
      if (threadIdx.x == 0) win_ctl[C::kCtlTile / 4] = next;
      __syncthreads();
      tile = rfl(win_ctl[C::kCtlTile / 4]);
    }
    if (tile < 0) break;  // the whole workgroup
    if (PERSIST && threadIdx.x == 0) next = win_next_slot(queue, home, tiles_per_xcd, order);  // in flight behind this tile
    PSH_DASSERT(tile < n_tiles);  // the order table names tiles of this launch's grid
    const int xt = (tile % tiles_x) * kTileX + lane;
    const int yt = row0 + (tile / tiles_x) * C::kTileY + static_cast<int>(threadIdx.x / kTileX) * kWinRows;
    const int x = min(xt, n - 1);
    float *out = out_base;
#ifndef PSH_WIN_MASKED_STORES
    const int out_plane_bytes = static_cast<int>(static_cast<unsigned>(__mul24(rows, n)) * 4u);  // (< 0xfffffff0: semilag_window_eligible)
#endif
    Window W;
    W.uv = static_cast<unsigned>(reinterpret_cast<size_t>((lds_void *)win_planes));  // 8-byte texels
    W.p = W.uv + 2u * C::kPlaneBytes;
    W.ctl = static_cast<unsigned>(reinterpret_cast<size_t>((lds_void *)win_ctl));
    W.ox = W.oy = 0;
    W.lo_x = W.lo_y = 0;
    W.hi_x = W.hi_y = 0;
    W.edge = 0;
    W.fill16 = ((n & 3) == 0 && (plane & 3) == 0 &&
                ((reinterpret_cast<uintptr_t>(vel) | reinterpret_cast<uintptr_t>(precip)) & 15) == 0)
                   ? 1
                   : 0;
    W.stats = stats;
    if (threadIdx.x < 3) win_ctl[C::kCtlFlag / 4 + threadIdx.x] = 0;

    int y[kWinRows], dxw[kWinRows], dy[kWinRows];
    f32x2 f[kWinRows], vi[kWinRows], s_uv[kWinRows];
    float sp[kWinRows];
    bool live[kWinRows];
    unsigned opix[kWinRows];
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      live[j] = xt < n && yt + j < row0 + rows;
      y[j] = min(yt + j, m - 1);
      opix[j] = static_cast<unsigned>(__mul24(y[j] - row0, n) + x) << 2;
#ifndef PSH_WIN_MASKED_STORES
      if (!live[j]) opix[j] = 0xfffffff0u;
#endif
      int px = x, py = y[j];
      float ifx = 0.f, ify = 0.f, ivx = 0.f, ivy = 0.f;
      if (resume) {
        split_displacement(disp[static_cast<size_t>(y[j]) * n + x], px, ifx);
        split_displacement(disp[plane + static_cast<size_t>(y[j]) * n + x], py, ify);
      }
      if (resume != 1) {  // (2: base positions of a custom grid in the buffer, the increment starts as the grid's velocity)
        const unsigned pix = static_cast<unsigned>(__mul24(y[j], n) + x) << 2;
        ivx = ld(F.u0, pix) * first_scale;  // first increment is NOT divided by n_iter (semilagrangian.py:202)
        ivy = ld(F.v0, pix) * first_scale;
      }
      dxw[j] = px;
      dy[j] = py;
      f[j] = f32x2{ifx, ify};
      vi[j] = f32x2{ivx, ivy};
      sp[j] = 0.f;
    }
    __syncthreads();
    int phase = 0;
    win_update<C, GEN>(F, W, phase, true, dxw, dy, 0.5f * vi[0].x, 0.5f * vi[0].y, move_scale, m, n);
    phase = 1;
    if (resume == 1) {
      win_sample<C, kVel, GEN>(F, W, dxw, dy, f, m, n, outval, s_uv, sp);
      const float s0 = scale[0];
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) vi[j] = s_uv[j] * s0;
    }
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) vi[j] = vi[j] * 0.5f;  // the increment is only ever used halved (exact)

    for (int t = 0; t < T; ++t) {
      const float s = scale[t];
      const float half_s = 0.5f * s;
      for (int k = 0; k < n_iter; ++k) {
        int mxw[kWinRows], my[kWinRows];
        f32x2 g[kWinRows];
#pragma unroll
        for (int j = 0; j < kWinRows; ++j) {
          mxw[j] = dxw[j];
          my[j] = dy[j];
          g[j] = f[j];
          retreat_xy(mxw[j], my[j], g[j], vi[j]);  // midpoint rule (:213), vi = Vi / 2
        }
        win_sample<C, kVel, GEN>(F, W, mxw, my, g, m, n, outval, s_uv, sp);
#pragma unroll
        for (int j = 0; j < kWinRows; ++j) retreat_xy(dxw[j], dy[j], f[j], s_uv[j] * s);
        if (k == n_iter - 1) {
          win_sample<C, kVel | kPrecip, GEN>(F, W, dxw, dy, f, m, n, outval, s_uv, sp);
        } else {
          win_sample<C, kVel, GEN>(F, W, dxw, dy, f, m, n, outval, s_uv, sp);
        }
#pragma unroll
        for (int j = 0; j < kWinRows; ++j) vi[j] = s_uv[j] * half_s;
      }
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        const float val = lost(f[j].x, f[j].y) ? lostval : sp[j];
#ifndef PSH_WIN_MASKED_STORES
        // one plane of the output as a buffer whose base moves (scalar adds) from lead step to lead step: the lane's
        // offset is all the address arithmetic, and the pixels outside the image or the band sit behind the buffer's
        // end, where the hardware drops the store - no exec mask, no branch
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), __builtin_amdgcn_make_buffer_rsrc(out, 0, out_plane_bytes, 0x00020000),
                                              static_cast<int>(opix[j]), 0, 2 /* nt */);
#else
        if (live[j]) __builtin_nontemporal_store(val, reinterpret_cast<float *>(reinterpret_cast<char *>(out) + opix[j]));
#endif
      }
      out += static_cast<size_t>(rows) * n;
      if (t + 1 < T) {
        win_update<C, GEN>(F, W, phase, false, dxw, dy, vi[0].x, vi[0].y, move_scale, m, n);
        phase = phase == 2 ? 0 : phase + 1;
      }
    }

    if (disp != nullptr) {
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        if (!live[j]) continue;
        disp[static_cast<size_t>(y[j]) * n + x] = static_cast<double>(sat_add(W.ox, dxw[j])) - static_cast<double>(x) + static_cast<double>(f[j].x);
        disp[plane + static_cast<size_t>(y[j]) * n + x] = static_cast<double>(sat_add(W.oy, dy[j])) - static_cast<double>(y[j]) + static_cast<double>(f[j].y);
      }
    }
    if (!PERSIST) return;
  }
  // persistent launch: the last workgroup out rewinds the cursors (the next launch on the stream starts from zero)
  if (PERSIST && threadIdx.x == 0) {
    if (__hip_atomic_fetch_add(queue + kNumXcd, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      for (int i = 0; i <= kNumXcd; ++i) __hip_atomic_store(queue + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// the window kernel reads the planes as they are: no packed copy of anything
static bool window_shape(int m, int n) { return n >= Win8::kW && m >= Win8::kH; }

bool semilag_window_eligible(const SemilagArgs &a) {
  // (a plane of the output below 0xfffffff0 bytes: the lanes without a pixel store to that offset, behind the buffer's end)
  return a.precip != nullptr && a.order == 1 && a.n_iter >= 1 && window_shape(a.m, a.n) &&
         static_cast<size_t>(a.m) * a.n * sizeof(float) < 0xfffffff0ull;
}

// ---- the order the tiles are started in (see win_next_slot) ----------------------------------------------
// One table per tile grid, built on the host and kept on the device.  Slot s belongs to XCD s % 8.
struct WinOrder {
  int tiles_x = 0, tiles_y = 0, tile_h = 0, mode = -1;
  int *dev = nullptr;
};
static WinOrder g_win_orders[8];
static int g_win_orders_next = 0;
static unsigned *g_win_queue = nullptr;  // kQueueRing x 16 words: the cursors of persistent launches
static unsigned g_win_queue_turn = 0;
constexpr int kQueueRing = 16;

// mode bit 0: XCD cells (else one band of tiles per XCD), bit 1: border rings first
static std::vector<int> win_order_table(int tiles_x, int tiles_y, int tile_h, int mode) {
  const int n_tiles = tiles_x * tiles_y, per = (n_tiles + kNumXcd - 1) / kNumXcd;
  const bool cells = (mode & 1) && tiles_x % kNumXcd == 0 && tiles_y % kNumXcd == 0;
  std::vector<int> table(static_cast<size_t>(per) * kNumXcd, -1);
  for (int k = 0; k < kNumXcd; ++k) {
    std::vector<int> mine;
    for (int l = 0; l < per; ++l) {
      int tile;
      if (cells) {
        const int cw = tiles_x / kNumXcd, ch = tiles_y / kNumXcd, per_cell = cw * ch;
        const int cy = l / per_cell, r = l - cy * per_cell;
        const int cx = (k + kNumXcd - cy) & (kNumXcd - 1);
        tile = (cy * ch + r / cw) * tiles_x + cx * cw + r % cw;
      } else {
        tile = k * per + l;
      }
      if (tile < n_tiles) mine.push_back(tile);
    }
    if (mode & 2) {
      // pixels between the tile and the nearest image border, in rings of 32 up to 192 (beyond: the interior)
      auto ring = [&](int tile) {
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int d = std::min(std::min(tx, tiles_x - 1 - tx) * kTileX, std::min(ty, tiles_y - 1 - ty) * tile_h);
        return std::min(d, 192) / 32;
      };
      std::stable_sort(mine.begin(), mine.end(), [&](int a, int b) { return ring(a) < ring(b); });
    }
    for (size_t l = 0; l < mine.size(); ++l) table[l * kNumXcd + k] = mine[l];
  }
  return table;
}

static const int *win_order(int tiles_x, int tiles_y, int tile_h, int mode, hipStream_t stream) {
  for (const WinOrder &o : g_win_orders)
    if (o.dev != nullptr && o.tiles_x == tiles_x && o.tiles_y == tiles_y && o.tile_h == tile_h && o.mode == mode) return o.dev;
  WinOrder &o = g_win_orders[g_win_orders_next];
  g_win_orders_next = (g_win_orders_next + 1) % 8;
  const std::vector<int> table = win_order_table(tiles_x, tiles_y, tile_h, mode);
  if (o.dev != nullptr) {
    (void)hipStreamSynchronize(stream);  // a launch in flight may still read the table this slot held
    (void)hipFree(o.dev);
    o.dev = nullptr;
  }
  if (hipMalloc(&o.dev, table.size() * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemcpy(o.dev, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(o.dev);
    o.dev = nullptr;
    return nullptr;
  }
  o.tiles_x = tiles_x;
  o.tiles_y = tiles_y;
  o.tile_h = tile_h;
  o.mode = mode;
  return o.dev;
}

static int env_int(const char *name, int fallback) {
  const char *e = std::getenv(name);
  return e ? std::atoi(e) : fallback;
}

template <class C>
static hipError_t launch_window(const SemilagArgs &a, hipStream_t stream) {
  const int tiles_x = (a.n + kTileX - 1) / kTileX;
  const int tiles_y = (a.rows + C::kTileY - 1) / C::kTileY;
  const int n_tiles = tiles_x * tiles_y;
  const int tiles_per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
  static const bool want_stats = std::getenv("PYSTEPS_HIP_SL_STATS") != nullptr;
  // room asked for ahead of the corner samples, in half increments per sub-step: 2.0 = the distance the last lead step
  // covered (+ 2 pixels); measured 2.0 / 2.2 / 2.5 / 3.0: 1.184 / 1.190 / 1.206 / 1.209 ms (development knob)
  static const float guard = std::getenv("PYSTEPS_HIP_SL_GUARD") ? static_cast<float>(std::atof(std::getenv("PYSTEPS_HIP_SL_GUARD"))) : 2.0f;
  // development knobs: PYSTEPS_HIP_SL_CELLS=0: one band of tiles per XCD; PYSTEPS_HIP_SL_RINGS=0: image order instead of
  // border rings first; PYSTEPS_HIP_SL_PERSIST=1: persistent workgroups on a tile queue (N > 1: N workgroups)
  static const int order_mode = (env_int("PYSTEPS_HIP_SL_CELLS", 1) ? 1 : 0) | (env_int("PYSTEPS_HIP_SL_RINGS", 1) ? 2 : 0);
  static const int persist = env_int("PYSTEPS_HIP_SL_PERSIST", 0);
  const int *order = win_order(tiles_x, tiles_y, C::kTileY, order_mode, stream);
  if (order == nullptr) return hipErrorOutOfMemory;
  unsigned *queue = nullptr;
  // workgroups the chip holds at once: kOcc waves per SIMD
  // (PYSTEPS_HIP_SL_PERSIST > 1: that many workgroups, whatever the grid - the tests' way to a queue on small images)
  const int resident = persist > 1 ? persist : psh::ctx().cu_count * (4 * C::kOcc / C::kWaves);
  int grid_x = tiles_per_xcd * kNumXcd;
  if (persist && (grid_x > resident || persist > 1)) {
    if (g_win_queue == nullptr) {
      if (hipMalloc(&g_win_queue, kQueueRing * 16 * sizeof(unsigned)) != hipSuccess) return hipErrorOutOfMemory;
      if (hipMemset(g_win_queue, 0, kQueueRing * 16 * sizeof(unsigned)) != hipSuccess) return hipErrorUnknown;
    }
    queue = g_win_queue + 16 * (g_win_queue_turn++ % kQueueRing);
    grid_x = resident;
  }
  const dim3 grid(grid_x), block(C::kThreads);
  if (want_stats) {
    if (g_win_stats == nullptr && hipMalloc(&g_win_stats, 4 * sizeof(unsigned long long)) != hipSuccess) g_win_stats = nullptr;
    if (g_win_stats != nullptr) (void)hipMemsetAsync(g_win_stats, 0, 4 * sizeof(unsigned long long), stream);
  }
#define PSH_WIN_LAUNCH(GEN, PERSIST)                                                                               \
  hipLaunchKernelGGL((semilag_window<C, GEN, PERSIST>), grid, block, 0, stream, a.precip, a.vel, a.out, a.disp, a.scale, \
                     a.first_scale, a.m, a.n, a.T, a.n_iter, a.resume, a.outval, a.row0, a.rows, a.bmode, tiles_x,      \
                     n_tiles, tiles_per_xcd, guard, order, queue, g_win_stats)
  if (a.bmode != 0) {
    if (queue != nullptr) PSH_WIN_LAUNCH(true, true); else PSH_WIN_LAUNCH(true, false);
  } else {
    if (queue != nullptr) PSH_WIN_LAUNCH(false, true); else PSH_WIN_LAUNCH(false, false);
  }
#undef PSH_WIN_LAUNCH
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess && want_stats && g_win_stats != nullptr) {
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(h, g_win_stats, sizeof(h), hipMemcpyDeviceToHost, stream) == hipSuccess &&
        hipStreamSynchronize(stream) == hipSuccess)
      std::fprintf(stderr, "semilag_window<%d waves x %d rows>: %dx%d T=%d: wave-passes through the window %llu, through the gathers %llu, "
                   "window fills (per wave) %llu (%d tiles on %d workgroups x %d lead steps)\n", C::kWaves, C::kRows, a.m, a.n, a.T,
                   h[0], h[1], h[2], n_tiles, grid_x, a.T);
  }
  return e;
}

template <int NPX, int MODE>
static hipError_t launch_variant(const SemilagArgs &a, hipStream_t stream) {
  constexpr int kWaves = kDirectWaves;
  const int tile_y = kWaves * NPX;
  const int tiles_x = (a.n + kTileX - 1) / kTileX;
  const int tiles_y = (a.rows + tile_y - 1) / tile_y;
  const int n_tiles = tiles_x * tiles_y;
  const int tiles_per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
  const dim3 grid(tiles_per_xcd * kNumXcd), block(kTileX * kWaves);
#define PSH_SL_LAUNCH(ORDER, HASP, GEN)                                                         \
  hipLaunchKernelGGL((semilag_fused<NPX, ORDER, HASP, MODE, GEN>), grid, block, 0, stream,      \
                     a.precip, a.vel, a.vel_packed, a.field_pairs, a.out, a.disp, a.scale, a.first_scale, a.m, a.n, \
                     a.T,                                                                                  \
                     a.n_iter, a.resume, a.outval, a.row0, a.rows, a.coef, a.minval, a.bmode, a.coef_pad | (a.spline_order << 8),  \
                     tiles_x, n_tiles, tiles_per_xcd)
  if (a.precip == nullptr) {
    PSH_SL_LAUNCH(1, false, false);
  } else if (a.order == 0) {
    if (a.bmode != 0) {
      PSH_SL_LAUNCH(0, true, true);
    } else {
      PSH_SL_LAUNCH(0, true, false);
    }
  } else if (a.order == 3) {
    if (a.bmode != 0) {
      PSH_SL_LAUNCH(3, true, true);
    } else {
      PSH_SL_LAUNCH(3, true, false);
    }
  } else if (a.bmode != 0) {
    PSH_SL_LAUNCH(1, true, true);
  } else {
    PSH_SL_LAUNCH(1, true, false);
  }
#undef PSH_SL_LAUNCH
  return hipGetLastError();
}

}  // namespace

// semilag_variant: 0 (default) = the workgroup-window kernel wherever it applies (bilinear resampling of a field,
// n_iter >= 1, images of at least 96 x 64 pixels, at least two sampling steps), the gather
// kernels elsewhere; 12 = the window kernel for every eligible call, however short; 7 = gather kernels only: velocity
// from a packed {u,v} plane and the field from a row-pair plane (the default of rounds 2 - 4); 5 = packed velocity
// only; 1 = one plane per component with DPP column sharing (what calls of fewer than 8 sampling steps take among the
// gather kernels).  All of them give bit-identical results (tests/test_semilag_gpu.py, tools/sl_bitcheck.py).
static int g_semilag_variant = [] {
  const char *e = std::getenv("PYSTEPS_HIP_SL_VARIANT");
  return e ? std::atoi(e) : 0;
}();

void set_semilag_variant(int v) { g_semilag_variant = v; }

// the first window costs one fill before anything is sampled: a call of a single sampling step keeps the gathers
// (4096^2, n_iter 1, T = 1 / 2 / 3 / 4 / 8: window 0.106 / 0.143 / 0.180 / 0.217 / 0.403 ms, gathers 0.100 / 0.153 / 0.205 /
// 0.262 / 0.597 ms - profiles/r05/d_window_default_knobs_short_calls.txt)
constexpr long long kWindowMinPasses = 2;

bool semilag_uses_window(const SemilagArgs &a) {
  if (!semilag_window_eligible(a)) return false;
  if (g_semilag_variant == 12) return true;
  return g_semilag_variant == 0 && static_cast<long long>(a.T) * a.n_iter >= kWindowMinPasses;
}

bool semilag_window_shape(int m, int n) { return window_shape(m, n); }

// which kernel a call of this shape takes under the current "semilag_variant" (16-byte aligned planes assumed):
// 12 the window kernel, 7 / 5 / 1 the gather modes (psh_semilag_kernel: bench.py labels its roofline with it)
int semilag_kernel_choice(int m, int n, int T, int n_iter, int order, bool has_field) {
  SemilagArgs a{};
  static float dummy[4] __attribute__((aligned(16)));
  a.precip = has_field ? dummy : nullptr;
  a.vel = dummy;
  a.m = m;
  a.n = n;
  a.T = T;
  a.n_iter = n_iter;
  a.order = order;
  if (semilag_uses_window(a)) return 12;
  if (semilag_wants_field_pairs(a)) return 7;
  if (semilag_wants_packed(a)) return 5;
  return 1;
}

hipError_t launch_semilag(const SemilagArgs &a, hipStream_t stream) {
  if (semilag_uses_window(a)) return launch_window<Win8>(a, stream);
  if (a.vel_packed != nullptr && a.field_pairs != nullptr && a.order == 1) return launch_variant<1, kModePacked2>(a, stream);
  if (a.vel_packed != nullptr) return launch_variant<1, kModePacked>(a, stream);
  return launch_variant<1, kModeDirect>(a, stream);
}

// The gather kernels sample the velocity from a packed {u,v} plane when the call is long enough to pay for the
// layout pass (one sweep over the planes, 0.04 ms at 4096^2, saves ~4 us per sampling pass of a 4096^2 step: from ~8
// sampling steps on) - unless the caller hands the packed plane over (psh_semilag_uv_dev).  Shorter calls - the
// single-step calls of a generic nowcast loop - take the one-plane-per-component kernel.  The window kernel needs
// no second layout of anything.
bool semilag_wants_packed(const SemilagArgs &a) {
  return !semilag_uses_window(a) && (g_semilag_variant == 0 || g_semilag_variant == 5 || g_semilag_variant == 7) &&
         static_cast<uint64_t>(a.m) * static_cast<uint64_t>(a.n) < (1ull << 29) &&
         static_cast<long long>(a.T) * (a.n_iter > 0 ? a.n_iter : 1) >= 8;
}
// ... and the field from a row-pair plane (one dwordx4 per sample); 5 = packed velocity only (two dwordx2 for the field)
bool semilag_wants_field_pairs(const SemilagArgs &a) {
  return (g_semilag_variant == 0 || g_semilag_variant == 7) && semilag_wants_packed(a) && a.precip != nullptr &&
         a.order == 1 && a.T >= 8;
}

hipError_t launch_pack_field_rows(const float *precip, float *pairs, int m, int n, hipStream_t stream) {
  hipLaunchKernelGGL(pack_field_rows, dim3((n + 1023) / 1024, m), dim3(256), 0, stream, precip, pairs, m, n);
  return hipGetLastError();
}

hipError_t launch_pack_velocity(const float *vel, float *uv, size_t plane, hipStream_t stream) {
  const size_t threads = (plane + 3) / 4;
  hipLaunchKernelGGL(pack_velocity, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, stream, vel, uv,
                     plane);
  return hipGetLastError();
}

}  // namespace psh
