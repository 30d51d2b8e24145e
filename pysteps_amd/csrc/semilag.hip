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
//  * The trajectory is carried as integer pixel position + fraction in [0,1) per axis.  Sub-pixel
//    weights therefore keep full fp32 precision however far the trajectory has
//    travelled, and the "advected from outside" test of map_coordinates
//    (coord < 0 or coord > len-1, strict) becomes an integer comparison.
//  * 64x4-pixel workgroups of 4 waves, one image row per wave, one pixel per lane (two rows
//    per thread were measured equal).  A wave reads 64 consecutive floats per tap row
//    (coalesced up to the sub-row shift).  The block index is remapped so that each XCD
//    (block b runs on XCD b % 8) owns one contiguous horizontal band of the image and its
//    private 4 MiB L2 sees all the halo reuse of that band.
//  * Waves whose 64 lanes all have their four taps strictly inside the image (almost all of
//    them) take a clamp-free path: buffer loads with one lane offset for every plane (+1 row
//    = scalar offset), the right-hand column of each lane's 2x2 footprint taken from lane
//    i+1 by DPP (v_cndmask_b32_dpp) unless the neighbour's trajectory sits elsewhere.
//  * No LDS (default variant), no MFMA: the gather footprint moves with D and there is no
//    dense contraction.  The kernel is bound by the CU's vector-memory pipeline.
#include <cstdlib>

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
// kernel flavours: direct gathers (one pixel per lane), LDS-staged tiles, direct gathers with
// two horizontally adjacent pixels per lane
// kModePacked: {u,v} interleaved velocity plane; kModePacked2: that plus the row-pair field plane
// kModeWave: packed velocity plane, every WAVE stages the bounding box of its own samples in LDS
enum : int { kModeDirect = 0, kModeStaged = 1, kModePairX = 2, kModePacked = 3, kModePacked2 = 4, kModeWave = 5 };
constexpr bool is_wave(int mode) { return mode == kModeWave; }
constexpr int kWavesPerBlock = 4;   // LDS-staged variants: 4 waves (rows) per workgroup
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

// ---- LDS-staged path ------------------------------------------------------------
// The L1 (TCP) is the unit the direct gathers saturate: a dword per
// lane costs it ~3x more per byte than a 16-byte-per-lane stream.  Here the workgroup
// (64x16 pixels, 4 rows per thread) first reduces the bounding box of all its sample
// positions (packed 16-bit min/max through the wave, 4-wave combine in LDS), fetches
// that box - tile plus the halo the displacement field needs - ONCE per plane with
// aligned, fully coalesced dwordx4 loads into LDS, and then takes the four taps of
// every pixel from LDS.  Boxes that touch the image border or exceed the LDS budget
// (strong deformation) fall back to the direct path, block-uniformly.
constexpr int kStageCap = 1792;  // floats per plane (7 KiB); 3 planes -> 21 KiB per workgroup

struct WaveFetch;
struct Stage {
  const WaveFetch *fetch;  // kModeWave: per-lane constants of the staged fetch
  unsigned lds;            // kModeWave: LDS byte address of the wave's region (scalar)
  unsigned lds_v, lds_p;   // ... and of its two planes, as opaque lane values (address arithmetic on VALU)
  float *buf;    // [3][kStageCap]
  int *red;      // [2][8] packed per-wave bounding boxes, double buffered
  int x0, y0;    // tile origin: positions are reduced relative to it in 16 bits
  int parity;
};

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int pk_min(int a, int b) {
  const short2v r = __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b));
  return __builtin_bit_cast(int, r);
}
__device__ __forceinline__ int pk_max(int a, int b) {
  const short2v r = __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b));
  return __builtin_bit_cast(int, r);
}
__device__ __forceinline__ int pk(int x, int y) {
  return (min(max(y, -32768), 32767) << 16) | (min(max(x, -32768), 32767) & 0xffff);
}

// Wave-wide reduction of an idempotent packed min/max in six DPP steps (pure VALU,
// no LDS round trips): xor-1 and xor-2 inside each quad, half-mirror and mirror
// inside each row of 16, then row_bcast15 / row_bcast31 across the four rows.
// Afterwards lane 63 holds the result for the whole wave.
template <bool IS_MIN>
__device__ __forceinline__ int wave_reduce_pk(int v) {
#define PSH_STEP(CTRL, ROWMASK)                                                            \
  {                                                                                         \
    const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false);            \
    v = IS_MIN ? pk_min(v, o) : pk_max(v, o);                                               \
  }
  PSH_STEP(0xB1, 0xf)   // quad_perm:[1,0,3,2]
  PSH_STEP(0x4E, 0xf)   // quad_perm:[2,3,0,1]
  PSH_STEP(0x141, 0xf)  // row_half_mirror
  PSH_STEP(0x140, 0xf)  // row_mirror
  PSH_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1 and 3
  PSH_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2 and 3
#undef PSH_STEP
  return v;
}

template <int NPX, int ORDER, bool WITH_P>
__device__ __forceinline__ bool sample_staged(const Fields &F, Stage &S, const int (&X)[NPX],
                                              const int (&Y)[NPX], const float (&fx)[NPX],
                                              const float (&fy)[NPX], int m, int n,
                                              float (&su)[NPX], float (&sv)[NPX],
                                              float (&sp)[NPX]) {
  // ---- bounding box of every sample position of the workgroup -------------------
  int lo = pk(X[0] - S.x0, Y[0] - S.y0), hi = lo;
#pragma unroll
  for (int j = 1; j < NPX; ++j) {
    const int q = pk(X[j] - S.x0, Y[j] - S.y0);
    lo = pk_min(lo, q);
    hi = pk_max(hi, q);
  }
  lo = wave_reduce_pk<true>(lo);
  hi = wave_reduce_pk<false>(hi);
  int *red = S.red + S.parity * 8;
  S.parity ^= 1;
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) {
    red[wave] = lo;
    red[4 + wave] = hi;
  }
  __syncthreads();
  lo = pk_min(pk_min(red[0], red[1]), pk_min(red[2], red[3]));
  hi = pk_max(pk_max(red[4], red[5]), pk_max(red[6], red[7]));
  const int lox = static_cast<short>(lo & 0xffff), loy = lo >> 16;
  const int hix = static_cast<short>(hi & 0xffff), hiy = hi >> 16;
  const int bx0 = lox + S.x0, by0 = loy + S.y0;
  const int rx0 = bx0 & ~3;                                  // 16-byte aligned row starts
  const int W = ((hix + S.x0 + 2 - rx0) + 3) & ~3;           // + right tap, rounded to 4
  const int H = hiy - loy + 2;                               // + lower tap
  const bool ok = lox > -32768 && loy > -32768 && hix < 32767 && hiy < 32767 && rx0 >= 0 &&
                  rx0 + W <= n && by0 >= 0 && by0 + H <= m && W * H <= kStageCap;
  if (!ok) return false;  // identical in every thread of the workgroup
  // ---- one coalesced fetch of the box per plane -----------------------------------
  const int W4 = W >> 2, items = H * W4;
  float *bu = S.buf, *bv = S.buf + kStageCap, *bp = S.buf + 2 * kStageCap;
  for (int it = threadIdx.x; it < items; it += kTileX * kWavesPerBlock) {
    const int row = it / W4, c4 = it - row * W4;
    const unsigned g = static_cast<unsigned>(__mul24(by0 + row, n) + rx0 + 4 * c4) << 2;
    const int l = row * W + 4 * c4;
    *reinterpret_cast<float4 *>(bu + l) =
        *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(F.u0) + g);
    *reinterpret_cast<float4 *>(bv + l) =
        *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(F.v0) + g);
    if (WITH_P)
      *reinterpret_cast<float4 *>(bp + l) =
          *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(F.p0) + g);
  }
  __syncthreads();
  // ---- taps from LDS ---------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    const int o = (Y[j] - by0) * W + (X[j] - rx0);
    const Weights w = make_weights(fx[j], fy[j]);
    su[j] = blend(w, bu[o], bu[o + 1], bu[o + W], bu[o + W + 1]);
    sv[j] = blend(w, bv[o], bv[o + 1], bv[o + W], bv[o + W + 1]);
    if (WITH_P) {
      if (ORDER == 1) {
        sp[j] = blend(w, bp[o], bp[o + 1], bp[o + W], bp[o + W + 1]);
      } else {
        sp[j] = bp[o + (fy[j] >= 0.5f ? W : 0) + (fx[j] >= 0.5f ? 1 : 0)];
      }
    }
  }
  return true;
}

// ---- per-wave LDS staging over the packed planes ------------------------------------------------
// The direct kernels ask the CU's vector memory pipeline for 16 B per lane and tap row although
// neighbouring lanes and rows want the same bytes again: 80 B per pixel and lead step for 36 B of
// unique data, and that pipeline (64 B/clk) is the unit the kernel saturates.
// Here a wave owns 64 x 4 pixels (4 rows per lane).  Per sampling pass it reduces the bounding box
// of its 256 sample positions (four interleaved v_min/v_max_i32_dpp chains, read back with
// v_readlane) and fetches the box - the pixels plus the halo the motion's shear needs - with
// `buffer_load_dwordx4 ... lds`: consecutive lanes carry consecutive 16-byte items, the data goes
// from the texture path straight into the wave's own LDS region (no VGPRs, no ds_write).  The box
// has a FIXED pitch of 72 pixels (36 velocity items / 18 field items per row) and at most 7 rows,
// so which (row, column) a lane fetches in the k-th instruction is a per-lane constant: the
// instruction's address is that constant + a scalar offset (the box origin), no address
// arithmetic at all, and the tap rows sit at immediate LDS offsets.  A 72 x 6 box of {u,v} pairs
// = 4 instructions instead of the 8 dwordx4 gathers of 4 pixels; the field (plain plane, 4
// pixels per item) 2 instead of 4.  Nothing is shared between waves: no barrier, and the hazard
// between a pass's LDS reads and the next pass's LDS-DMA writes is a data dependence (the next box is
// a function of the values read).  A box that does not fit (shear of more than ~6 pixels across
// the 64 x 4 patch, a lost trajectory parked far away) falls back to the direct gathers, wave by
// wave and pass by pass; both paths blend the same values in the same order.
constexpr int kBoxPitch = 72;                           // pixels per box row
constexpr int kBoxRows = 7;
constexpr int kWaveVelItems = 256;                      // 16-byte items: kBoxRows * 36 = 252
constexpr int kWaveFieldItems = 128;                    // kBoxRows * 18 = 126
constexpr int kWaveLdsFloats = (kWaveVelItems + kWaveFieldItems) * 4;  // 6 KiB per wave

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;
typedef __attribute__((address_space(3))) float lds_f32;

// per-lane constants of the staged fetch: byte offset (relative to the box origin) of the item
// this lane fetches in the k-th instruction
struct WaveFetch {
  unsigned vel[kWaveVelItems / 64];
  unsigned field[kWaveFieldItems / 64];
};

__device__ __forceinline__ WaveFetch make_wave_fetch(int n) {
  WaveFetch w;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < kWaveVelItems / 64; ++k) {
    const int item = lane + 64 * k, row = item / (kBoxPitch / 2), col = item - row * (kBoxPitch / 2);
    w.vel[k] = static_cast<unsigned>(row * n + 2 * col) << 3;
  }
#pragma unroll
  for (int k = 0; k < kWaveFieldItems / 64; ++k) {
    const int item = lane + 64 * k, row = item / (kBoxPitch / 4), col = item - row * (kBoxPitch / 4);
    w.field[k] = static_cast<unsigned>(row * n + 4 * col) << 2;
  }
  return w;
}

// The box is CHOSEN from the four corner samples of the 64 x 4 patch (eight v_readlane + scalar
// min / max: exact when the motion is affine across the patch) and VERIFIED for every sample by the
// two differences the LDS address needs anyway, compared against the box while the fetch is in
// flight.  (A first version reduced the exact bounding box with four interleaved 6-step
// v_min/v_max_i32_dpp chains per pass: 98 VALU instructions per pixel and lead step instead of 68,
// 88 % VALU-bound - profiles/r03/i_semilag_wave_pmc.csv.)
// 0: sampled from LDS; 1: every tap inside the image, but the box does not fit; 2: border wave
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

template <int NPX, bool WITH_P>
__device__ __forceinline__ int sample_wave_staged(const Fields &F, Stage &S, const WaveFetch &wf, const int (&X)[NPX],
                                                   const int (&Y)[NPX], const float (&fx)[NPX],
                                                   const float (&fy)[NPX], int m, int n, float (&su)[NPX],
                                                   float (&sv)[NPX], float (&sp)[NPX]) {
  const int xa = __builtin_amdgcn_readlane(X[0], 0), xb = __builtin_amdgcn_readlane(X[0], 63);
  const int xc = __builtin_amdgcn_readlane(X[NPX - 1], 0), xd = __builtin_amdgcn_readlane(X[NPX - 1], 63);
  const int ya = __builtin_amdgcn_readlane(Y[0], 0), yb = __builtin_amdgcn_readlane(Y[0], 63);
  const int yc = __builtin_amdgcn_readlane(Y[NPX - 1], 0), yd = __builtin_amdgcn_readlane(Y[NPX - 1], 63);
  const int rx = smax((smin(smin(xa, xb), smin(xc, xd)) - 1) & ~3, 0);  // one pixel of slack, 16-byte aligned
  const int by0 = smax(smin(smin(ya, yb), smin(yc, yd)), 0);
  const int H = smin(smax(smax(ya, yb), smax(yc, yd)) - by0 + 2, kBoxRows);  // lower tap row included
  // what a sample's offset inside the box may be: all four taps inside the box and inside the image
  const int dx_max = smin(kBoxPitch - 2, n - 2 - rx), dy_max = smin(H - 2, m - 2 - by0);
  const int lane = threadIdx.x & 63;
  const int items_v = H * (kBoxPitch / 2), items_p = H * (kBoxPitch / 4);
  const int org = by0 * n + rx;
  const bool fetch = dx_max >= 0 && dy_max >= 0;  // a patch outside the image fetches nothing
  if (fetch) {
#pragma unroll
    for (int k = 0; k < kWaveVelItems / 64; ++k) {
      if (k * 64 >= items_v) break;
      if (lane + 64 * k < items_v)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(F.ruv, (lds_void *)(size_t)(S.lds + 1024u * k), 16,
                                                 static_cast<int>(wf.vel[k]), org << 3, 0, 0);
    }
    if (WITH_P) {
#pragma unroll
      for (int k = 0; k < kWaveFieldItems / 64; ++k) {
        if (k * 64 >= items_p) break;
        if (lane + 64 * k < items_p)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(F.rp, (lds_void *)(size_t)(S.lds + kWaveVelItems * 16u + 1024u * k), 16,
                                                   static_cast<int>(wf.field[k]), org << 2, 0, 0);
      }
    }
  }
  int o[NPX];
  unsigned dx_hi = 0, dy_hi = 0;  // as unsigned numbers: a negative difference is a huge one
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    const int dx = X[j] - rx, dy = Y[j] - by0;
    dx_hi = max(dx_hi, static_cast<unsigned>(dx));
    dy_hi = max(dy_hi, static_cast<unsigned>(dy));
    o[j] = __mul24(dy, kBoxPitch) + dx;
  }
  const bool ok = dx_hi <= static_cast<unsigned>(dx_max) && dy_hi <= static_cast<unsigned>(dy_max);
  const bool all_ok = fetch && __builtin_amdgcn_ballot_w64(ok) == __builtin_amdgcn_ballot_w64(true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // also on the way out: the region is fetched into again
  if (!all_ok) {
    bool inside = true;
#pragma unroll
    for (int j = 0; j < NPX; ++j) inside = inside && wave_all_interior(X[j], Y[j], m, n);
    return inside ? 1 : 2;
  }
  // LDS byte addresses: one shift-add per plane on top of the item index; the tap rows are immediates
  const unsigned vbase = S.lds_v, pbase = S.lds_p;
#pragma unroll
  for (int j = 0; j < NPX; ++j) {
    const lds_f32x2 *lv = (const lds_f32x2 *)(size_t)(vbase + (static_cast<unsigned>(o[j]) << 3));
    const lds_f32 *lp = (const lds_f32 *)(size_t)(pbase + (static_cast<unsigned>(o[j]) << 2));
    const f32x2 t0 = lv[0], t1 = lv[1], b0 = lv[kBoxPitch], b1 = lv[kBoxPitch + 1];
    const Weights w = make_weights(fx[j], fy[j]);
    f32x2 acc = t0 * w.w00;  // the order of sample_interior_packed
    acc = __builtin_elementwise_fma(f32x2{w.w01, w.w01}, t1, acc);
    acc = __builtin_elementwise_fma(f32x2{w.w10, w.w10}, b0, acc);
    acc = __builtin_elementwise_fma(f32x2{w.w11, w.w11}, b1, acc);
    su[j] = acc.x;
    sv[j] = acc.y;
    if (WITH_P) sp[j] = blend(w, lp[0], lp[1], lp[kBoxPitch], lp[kBoxPitch + 1]);
  }
  return 0;
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
  if (GEN) return sample_precip_cubic_mode(F.coef, F.p0, X, Y, fx, fy, m, n, F.minval, outval, F.bmode, F.cpad);
  return sample_precip_cubic(F.coef, F.p0, X, Y, fx, fy, m, n, F.minval);
}

// What to sample at the NPX positions of a thread
enum : int { kVel = 1, kPrecip = 2 };

template <int NPX, int ORDER, int WHAT, int MODE, bool GEN>
__device__ __forceinline__ void sample_at(const Fields &F, Stage &S, const int (&X)[NPX],
                                          const int (&Y)[NPX], const float (&fx)[NPX],
                                          const float (&fy)[NPX], int m, int n, float outval,
                                          float (&su)[NPX], float (&sv)[NPX], float (&sp)[NPX]) {
  constexpr bool kWithP = (WHAT & kPrecip) != 0;
  if (MODE == kModeStaged) {
    // every thread of the workgroup reaches this call (uniform loop structure)
    if (sample_staged<NPX, ORDER, kWithP>(F, S, X, Y, fx, fy, m, n, su, sv, sp)) return;
    // rare (border tiles, extreme deformation): plain clamped gathers, pixel by pixel
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      if (WHAT & kVel) sample_velocity_border(F, X[j], Y[j], fx[j], fy[j], m, n, su[j], sv[j]);
      if (kWithP) sp[j] = sample_precip_off_fast<ORDER, GEN>(F.p0, X[j], Y[j], fx[j], fy[j], m, n, outval, F.bmode);
    }
    return;
  }
  bool inside = true, staged = false;
  if (is_wave(MODE)) {
    // the bounding box of the wave's samples answers both questions
    const int st = sample_wave_staged<NPX, kWithP && ORDER == 1>(F, S, *S.fetch, X, Y, fx, fy, m, n, su, sv, sp);
    staged = st == 0;
    inside = st != 2;
  } else {
#pragma unroll
    for (int j = 0; j < NPX; ++j) inside = inside && wave_all_interior(X[j], Y[j], m, n);
  }
  // wave-uniform branch: interior waves (almost all of them) skip every clamp
  if (inside) {
    if (staged) {
    } else if (MODE == kModePacked || MODE == kModePacked2 || is_wave(MODE)) {
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
template <int MODE>
constexpr int waves_of() {
  return (MODE == kModeDirect || MODE == kModePacked || MODE == kModePacked2) ? kDirectWaves : kWavesPerBlock;  // kModeWave: 4
}

// kModeWave: 128 registers at most, so that four workgroups (16 waves) share a CU
template <int MODE>
constexpr int min_waves_per_simd() {
  return is_wave(MODE) ? 4 : 1;
}

template <int NPX, int ORDER, bool HAS_PRECIP, int MODE, bool GEN>
__global__ __launch_bounds__(kTileX *waves_of<MODE>(), min_waves_per_simd<MODE>()) void semilag_fused(
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
  constexpr int kWaves = waves_of<MODE>();
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
  constexpr bool kPackedVel = MODE == kModePacked || MODE == kModePacked2 || is_wave(MODE);
  F.ruv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(kPackedVel ? vel_packed : vel), 0, 2 * plane_bytes,
                                            0x00020000);
  F.rpp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(MODE == kModePacked2 ? field_pairs : vel), 0,
                                            2 * plane_bytes, 0x00020000);
  F.row_bytes = n * static_cast<int>(sizeof(float));
  F.coef = coef;
  F.cpad = coef_pad;
  F.minval = minval;
  F.bmode = bmode;

  __shared__ __attribute__((aligned(16))) float
      stage_buf[MODE == kModeStaged ? 3 * kStageCap : (is_wave(MODE) ? kWavesPerBlock * kWaveLdsFloats : 4)];
  __shared__ __attribute__((aligned(16))) int stage_red[16];
  Stage S;
  S.buf = stage_buf + (is_wave(MODE) ? (threadIdx.x >> 6) * kWaveLdsFloats : 0);
  S.red = stage_red;
  S.x0 = (tile % tiles_x) * kTileX;
  S.y0 = row0 + (tile / tiles_x) * (kWaves * NPX);
  S.parity = 0;
  WaveFetch wave_fetch;
  if (is_wave(MODE)) wave_fetch = make_wave_fetch(n);
  S.fetch = &wave_fetch;
  S.lds = S.lds_v = S.lds_p = 0;
  if (MODE == kModeWave) {
    S.lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_void *)S.buf)));
    S.lds_v = S.lds;
    S.lds_p = S.lds + kWaveVelItems * 16u;
    // kept apart from the compiler's constant folding: "base + 4096" does not fit the 8-bit offsets of
    // ds_read2 and would be re-added per tap row
    asm volatile("" : "+v"(S.lds_v), "+v"(S.lds_p));
  }

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
    sample_at<NPX, ORDER, kVel, MODE, GEN>(F, S, px, py, fx, fy, m, n, outval, su, sv, sp);
    const float s0 = scale[0];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
      vix[j] = su[j] * s0;
      viy[j] = sv[j] * s0;
    }
  } else {
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
        sample_at<NPX, ORDER, kVel, MODE, GEN>(F, S, mx, my, gx, gy, m, n, outval, su, sv, sp);
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
          retreat(px[j], fx[j], su[j] * s);
          retreat(py[j], fy[j], sv[j] * s);
        }
        if (HAS_PRECIP && k == n_iter - 1) {
          sample_at<NPX, ORDER, kVel | kPrecip, MODE, GEN>(F, S, px, py, fx, fy, m, n, outval, su, sv, sp);
        } else {
          sample_at<NPX, ORDER, kVel, MODE, GEN>(F, S, px, py, fx, fy, m, n, outval, su, sv, sp);
        }
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
          vix[j] = su[j] * half_s;
          viy[j] = sv[j] * half_s;
        }
      }
    } else {
      if (t > 0 || resume) {
        sample_at<NPX, ORDER, kVel, MODE, GEN>(F, S, px, py, fx, fy, m, n, outval, su, sv, sp);
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
      // kModeWave: the plane of this lead time as a buffer - scalar descriptor + 32-bit lane offset;
      // four 64-bit lane addresses carried through the loop cost 8 registers this variant does not have
      const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
          out, 0, is_wave(MODE) ? rows * n * static_cast<int>(sizeof(float)) : 0, 0x00020000);
#pragma unroll
      for (int j = 0; j < NPX; ++j) {
        // Non-finite velocities (allow_nonfinite_values, semilagrangian.py:106-137): a trajectory that
        // sampled one carries a NaN fraction from then on (v_cvt_flr(NaN) = 0 keeps the integer part in
        // range, every later sample is NaN).  map_coordinates answers a NaN coordinate with cval in
        // the "constant" mode (and the folding modes), with NaN where it interpolates across it.
        sp[j] = lost(fx[j], fy[j]) ? lostval : sp[j];
        // streamed once, never re-read: keep the output out of the L2 ways the input planes live in
        if (is_wave(MODE)) {
          if (live[j])
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sp[j]), rout, static_cast<int>(opix[j]), 0, 2 /* nt */);
        } else if (live[j]) {
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
          static_cast<double>(px[j] - x) + static_cast<double>(fx[j]);
      disp[plane + static_cast<size_t>(y[j]) * n + x] =
          static_cast<double>(py[j] - y[j]) + static_cast<double>(fy[j]);
    }
  }
}

template <int NPX, int MODE>
static hipError_t launch_variant(const SemilagArgs &a, hipStream_t stream) {
  constexpr int kWaves = waves_of<MODE>();
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
                     a.n_iter, a.resume, a.outval, a.row0, a.rows, a.coef, a.minval, a.bmode, a.coef_pad,  \
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
    if constexpr (MODE != kModeStaged) {
      if (a.bmode != 0) {
        PSH_SL_LAUNCH(3, true, true);
      } else {
        PSH_SL_LAUNCH(3, true, false);
      }
    } else {
      return hipErrorInvalidValue;  // the staged variants are built for order 0/1
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

// 0 = one pixel per lane, direct gathers + DPP column sharing (default); 4 / 2 = LDS-staged
// tiles with 4 / 2 rows per thread (measured equal at 4096^2 x 24, DESIGN.md 3.1: staging cuts
// the L1 traffic but adds two barriers per sampling pass); 3 = three pixels per lane with
// dwordx4 gathers (semilag_wide.hip: faster in near-uniform motion, slower once most waves
// carry a trajectory split).
static int g_semilag_variant = [] {
  const char *e = std::getenv("PYSTEPS_HIP_SL_VARIANT");
  return e ? std::atoi(e) : 0;
}();

void set_semilag_variant(int v) { g_semilag_variant = v; }

hipError_t launch_semilag(const SemilagArgs &a, hipStream_t stream) {
  // LDS staging needs 16-byte aligned rows (n % 4 == 0)
  const bool aligned = (a.n % 4 == 0) && (reinterpret_cast<uintptr_t>(a.vel) % 16 == 0) &&
                       (a.precip == nullptr || reinterpret_cast<uintptr_t>(a.precip) % 16 == 0);
  if (g_semilag_variant == 3 && semilag_wide_eligible(a)) return launch_semilag_wide(a, stream);
  if ((g_semilag_variant == 2 || g_semilag_variant == 4) && a.bmode == 0 && aligned && a.n >= 64 && a.m >= 16 && a.order != 3) {
    if (g_semilag_variant == 2) return launch_variant<2, kModeStaged>(a, stream);
    return launch_variant<4, kModeStaged>(a, stream);
  }
  // one or two rows per thread measured equal (1.45 ms at 4096^2 x 24): the kernel is not short of
  // loads in flight
  if (a.vel_packed != nullptr && a.field_pairs != nullptr && a.order == 1) {
    if (g_semilag_variant == 6) return launch_variant<2, kModePacked2>(a, stream);  // two rows per lane (experiment)
    return launch_variant<1, kModePacked2>(a, stream);
  }
  // variant 8: per-wave LDS staging of the packed velocity plane and the plain field plane
  if (a.vel_packed != nullptr && g_semilag_variant == 8 && a.order == 1 && aligned &&
      reinterpret_cast<uintptr_t>(a.vel_packed) % 16 == 0)
    return launch_variant<4, kModeWave>(a, stream);
  if (a.vel_packed != nullptr) return launch_variant<1, kModePacked>(a, stream);
  return launch_variant<1, kModeDirect>(a, stream);
}

// variant 0 (default) samples the velocity from a packed {u,v} plane when the caller provides one;
// variant 1 = the one-plane-per-component kernel with DPP column sharing (round 1 default)
// The layout passes cost one sweep over the planes each (0.04 ms at 4096^2) and save ~4 us per
// sampling pass of a 4096^2 step: they pay off from ~8 sampling steps on.  Shorter calls - the
// single-step calls of a generic nowcast loop - take the planar kernel (bit-identical results).
bool semilag_wants_packed(const SemilagArgs &a) {
  return (g_semilag_variant == 0 || g_semilag_variant == 5 || g_semilag_variant == 6 || g_semilag_variant == 8) &&
         static_cast<uint64_t>(a.m) * static_cast<uint64_t>(a.n) < (1ull << 29) &&
         static_cast<long long>(a.T) * (a.n_iter > 0 ? a.n_iter : 1) >= 8;
}
// variant 0 also samples the field from a row-pair plane (one dwordx4 per sample); 5 = packed
// velocity only (two dwordx2 for the field), kept for comparison; variant 8 stages the plain field
// plane through LDS and needs no second copy of it
bool semilag_wants_field_pairs(const SemilagArgs &a) {
  return (g_semilag_variant == 0 || g_semilag_variant == 6) && semilag_wants_packed(a) && a.precip != nullptr &&
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
