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
#include <cstdio>
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
// kernel flavours: kModeDirect: one plane per component, DPP column sharing (short calls, no packed plane);
// kModePacked: {u,v} interleaved velocity plane; kModePacked2: that plus the row-pair field plane
// kModeWave: packed velocity plane, every WAVE stages the bounding box of its own samples in LDS
enum : int { kModeDirect = 0, kModePacked = 3, kModePacked2 = 4, kModeWave = 5 };
constexpr bool is_wave(int mode) { return mode == kModeWave; }
constexpr int kWavesPerBlock = 4;   // kModeWave: 4 waves per workgroup
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

struct WaveFetch;
struct Stage {
  const WaveFetch *fetch;  // kModeWave: per-lane constants of the staged fetch
  unsigned lds;            // kModeWave: LDS byte address of the wave's region (scalar)
  unsigned lds_v, lds_p;   // ... and of its two planes, as opaque lane values (address arithmetic on VALU)
};

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

  __shared__ __attribute__((aligned(16))) float stage_buf[is_wave(MODE) ? kWavesPerBlock * kWaveLdsFloats : 4];
  Stage S;
  float *wave_buf = stage_buf + (is_wave(MODE) ? (threadIdx.x >> 6) * kWaveLdsFloats : 0);
  WaveFetch wave_fetch;
  if (is_wave(MODE)) wave_fetch = make_wave_fetch(n);
  S.fetch = &wave_fetch;
  S.lds = S.lds_v = S.lds_p = 0;
  if (MODE == kModeWave) {
    S.lds = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>((lds_void *)wave_buf)));
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

// ---- workgroup window (variant 9) ----------------------------------------------------------------
// The gather kernels above are bound by the CU's vector-memory pipeline: five wave64 dwordx4 gathers per
// pixel and lead step return 80 B per lane through a 64 B/clk path, whatever the caches
// hold.  Here a workgroup of four waves owns a 64 x 16 tile (four rows per lane) and keeps a WINDOW of the
// {u,v} plane and of the field plane - 96 x 40 texels around the tile's current sample positions - in LDS
// ACROSS lead steps: a sampling pass is LDS reads (4 x 8 B + 4 x 4 B per sample, conflict-free for
// neighbouring pixels) and arithmetic, nothing else.  The trajectory is carried relative to the window
// (pre-scaled column offset, row offset), so the two unsigned compares that prove "all four taps inside
// the window" replace the image-interior test and the LDS address is one multiply-add.
//  * A wave whose samples are not all inside the window (border of the image, extreme deformation, a
//    lost trajectory) takes that pass through the gathers of the packed kernel - wave by wave, pass by
//    pass, same values, same blend order: results are bit-identical to every other variant.
//  * Once per lead step the waves agree (one s_barrier) on whether the window has to move: a wave asks
//    for it when the corner samples of its patch come closer to the window's edge than the distance the
//    next step covers.  The new window is placed with its slack AHEAD of the motion (it then lasts
//    slack / speed lead steps: ~5 at 6 px per step), filled by coalesced dwordx4 loads + ds_write_b128,
//    and the lanes re-base their offsets.  Window traffic per pixel and lead step: ~0.6 vector-memory
//    instructions instead of 5.
constexpr int kWinRows = 4;   // image rows per lane
// WAVES waves per workgroup (a 64 x 4 WAVES tile), window of WW x WH texels (WW a multiple of 4)
template <int WAVES, int WW, int WH, bool RAW = false>
struct WinCfg {
  static constexpr int kWaves = WAVES, kW = WW, kH = WH;
  static constexpr bool kRaw = RAW;  // taps as single ds_read_b64 / ds_read_b32 (2 LDS cycles each; the read2 forms take 8 / 4)
  static constexpr int kTileY = kWinRows * WAVES;
  static constexpr unsigned kPitch8 = WW * 8u;        // bytes per window row of {u,v} pairs
  static constexpr int kItemsUV = WH * (WW / 2);      // 16-byte items of the {u,v} window
  static constexpr int kItemsP = WH * (WW / 4);       // ... of the field window
  static constexpr int kThreads = kTileX * WAVES;
  static constexpr unsigned kCtlVel = 16u * WAVES, kCtlFlag = kCtlVel + 8u;  // byte offsets in the control block
  static constexpr int kCtlWords = 4 * WAVES + 2 + 3 + 3;
};
using Win4 = WinCfg<4, 96, 40>;  // 45 KiB of LDS: three workgroups (12 waves) per CU
using Win8 = WinCfg<8, 96, 64>;  // 72 KiB: two workgroups (16 waves) per CU, the window lasts about twice as long
using Win8Raw = WinCfg<8, 96, 64, true>;

typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) int lds_int;

struct Window {
  unsigned uv, p;  // LDS byte addresses of the two planes
  unsigned ctl;    // ... of the control words: box[waves][4], vel[2], flag[3]
  int ox, oy;      // image position of the window's first texel (uniform over the workgroup)
  unsigned long long *stats;  // debug counters (nullptr): [0] passes through the window, [1] through the gathers, [2] fills
};

// LDS traffic only: the nontemporal output stores of the lead step stay in flight across the barrier
__device__ __forceinline__ void win_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ void win_count(const Window &W, int which) {
  if (W.stats != nullptr && (threadIdx.x & 63) == 0) atomicAdd(W.stats + which, 1ull);
}

template <class C, int WHAT, bool GEN>
__device__ __forceinline__ void win_sample(const Fields &F, const Window &W, const int (&dx8)[kWinRows],
                                           const int (&dy)[kWinRows], const float (&fx)[kWinRows],
                                           const float (&fy)[kWinRows], int m, int n, float outval,
                                           float (&su)[kWinRows], float (&sv)[kWinRows], float (&sp)[kWinRows]) {
  constexpr bool kWithP = (WHAT & kPrecip) != 0;
  // all four taps of all four samples inside the window: one unsigned maximum per axis (a negative offset is a huge one)
  unsigned mx = static_cast<unsigned>(dx8[0]), my = static_cast<unsigned>(dy[0]);
#pragma unroll
  for (int j = 1; j < kWinRows; ++j) {
    mx = max(mx, static_cast<unsigned>(dx8[j]));
    my = max(my, static_cast<unsigned>(dy[j]));
  }
  const bool ok = mx <= (C::kW - 2) * 8u && my <= static_cast<unsigned>(C::kH - 2);
  if (__builtin_amdgcn_ballot_w64(ok) == __builtin_amdgcn_ballot_w64(true)) {
    win_count(W, 0);
    // every read is issued before the first blend
    f32x2 t0[kWinRows], t1[kWinRows], b0[kWinRows], b1[kWinRows];
    float pa[kWinRows], pb[kWinRows], pc[kWinRows], pd[kWinRows];
    if (C::kRaw) {
      // one ds_read_b64 per {u,v} pair and one ds_read_b32 per field value: the LDS serves those at 2 cycles
      // per wave-instruction, the two-address forms the compiler would merge them into at 8 and 4
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        const unsigned a = __umul24(static_cast<unsigned>(dy[j]), C::kPitch8) + static_cast<unsigned>(dx8[j]) + W.uv;
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:8\n\tds_read_b64 %2, %4 offset:%c5\n\t"
                     "ds_read_b64 %3, %4 offset:%c6"
                     : "=&v"(t0[j]), "=&v"(t1[j]), "=&v"(b0[j]), "=&v"(b1[j])
                     : "v"(a), "i"(C::kPitch8), "i"(C::kPitch8 + 8));
        if (kWithP) {
          const unsigned ap = ((a - W.uv) >> 1) + W.p;
          asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:%c5\n\t"
                       "ds_read_b32 %3, %4 offset:%c6"
                       : "=&v"(pa[j]), "=&v"(pb[j]), "=&v"(pc[j]), "=&v"(pd[j])
                       : "v"(ap), "i"(C::kPitch8 / 2), "i"(C::kPitch8 / 2 + 4));
        }
      }
      // the reads above are invisible to the compiler's counters: wait for them, and tie every destination to the wait
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        if (kWithP) {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0[j]), "+v"(t1[j]), "+v"(b0[j]), "+v"(b1[j]), "+v"(pa[j]), "+v"(pb[j]), "+v"(pc[j]), "+v"(pd[j]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0[j]), "+v"(t1[j]), "+v"(b0[j]), "+v"(b1[j]));
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      const unsigned a = __umul24(static_cast<unsigned>(dy[j]), C::kPitch8) + static_cast<unsigned>(dx8[j]);
      const lds_f32x2 *q = (const lds_f32x2 *)(size_t)(W.uv + a);
      t0[j] = q[0], t1[j] = q[1], b0[j] = q[C::kW], b1[j] = q[C::kW + 1];
      if (kWithP) {
        const lds_f32 *r = (const lds_f32 *)(size_t)(W.p + (a >> 1));
        pa[j] = r[0], pb[j] = r[1], pc[j] = r[C::kW], pd[j] = r[C::kW + 1];
      }
    }
    }
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      const Weights w = make_weights(fx[j], fy[j]);
      f32x2 acc = t0[j] * w.w00;  // the order of sample_interior_packed
      acc = __builtin_elementwise_fma(f32x2{w.w01, w.w01}, t1[j], acc);
      acc = __builtin_elementwise_fma(f32x2{w.w10, w.w10}, b0[j], acc);
      acc = __builtin_elementwise_fma(f32x2{w.w11, w.w11}, b1[j], acc);
      su[j] = acc.x;
      sv[j] = acc.y;
      if (kWithP) sp[j] = blend(w, pa[j], pb[j], pc[j], pd[j]);
    }
  } else {
    win_count(W, 1);
    int X[kWinRows], Y[kWinRows];
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      X[j] = W.ox + (dx8[j] >> 3);
      Y[j] = W.oy + dy[j];
    }
    Stage S;
    S.fetch = nullptr;
    S.lds = S.lds_v = S.lds_p = 0;
    sample_at<kWinRows, 1, WHAT, kModePacked, GEN>(F, S, X, Y, fx, fy, m, n, outval, su, sv, sp);
  }
}

// Once per lead step, every wave of the workgroup: publish the box of the patch's corner samples, ask for
// a new window if they are about to leave this one, agree at a barrier, and if anybody asked: place the
// window ahead of the motion, re-base the offsets, fill it.  `phase` cycles through three flag words so
// that clearing the next one never races with a wave that still has to read it.
template <class C>
__device__ __forceinline__ void win_update(const Fields &F, Window &W, int phase, bool force, int (&dx8)[kWinRows],
                                           int (&dy)[kWinRows], float vx_lane, float vy_lane, float move_scale, int m,
                                           int n) {
  const int lane = threadIdx.x & 63, wave = rfl(static_cast<int>(threadIdx.x >> 6));
  lds_int *ctl = (lds_int *)(size_t)W.ctl;
  const int xa = __builtin_amdgcn_readlane(dx8[0], 0), xb = __builtin_amdgcn_readlane(dx8[0], 63);
  const int xc = __builtin_amdgcn_readlane(dx8[kWinRows - 1], 0), xd = __builtin_amdgcn_readlane(dx8[kWinRows - 1], 63);
  const int ya = __builtin_amdgcn_readlane(dy[0], 0), yb = __builtin_amdgcn_readlane(dy[0], 63);
  const int yc = __builtin_amdgcn_readlane(dy[kWinRows - 1], 0), yd = __builtin_amdgcn_readlane(dy[kWinRows - 1], 63);
  const int lo8 = smin(smin(xa, xb), smin(xc, xd)), hi8 = smax(smax(xa, xb), smax(xc, xd));
  const int loy = smin(smin(ya, yb), smin(yc, yd)), hiy = smax(smax(ya, yb), smax(yc, yd));
  // the wave's direction of travel and the distance one lead step covers (half increment of the first pixel;
  // a lost trajectory - NaN - asks for nothing)
  const float vx = __builtin_amdgcn_readfirstlane(vx_lane), vy = __builtin_amdgcn_readfirstlane(vy_lane);
  const float mx = fabsf(vx) < 64.f ? fabsf(vx) * move_scale + 2.f : 2.f, my = fabsf(vy) < 64.f ? fabsf(vy) * move_scale + 2.f : 2.f;
  const int gx = rfl(static_cast<int>(mx)), gy = rfl(static_cast<int>(my));
  // a positive velocity moves the samples towards lower coordinates (retreat)
  const int need_lx = vx > 0.f ? gx : 1, need_hx = vx > 0.f ? 1 : gx;
  const int need_ly = vy > 0.f ? gy : 1, need_hy = vy > 0.f ? 1 : gy;
  const bool near = force || lo8 < need_lx * 8 || hi8 > (C::kW - 2 - need_hx) * 8 || loy < need_ly || hiy > C::kH - 2 - need_hy;
  if (lane == 0) {
    if (near) ctl[C::kCtlFlag / 4 + phase] = 1;
    ctl[wave * 4 + 0] = lo8;
    ctl[wave * 4 + 1] = hi8;
    ctl[wave * 4 + 2] = loy;
    ctl[wave * 4 + 3] = hiy;
    if (wave == 0) {
      ctl[C::kCtlVel / 4 + 0] = __float_as_int(vx);
      ctl[C::kCtlVel / 4 + 1] = __float_as_int(vy);
      ctl[C::kCtlFlag / 4 + (phase == 2 ? 0 : phase + 1)] = 0;
    }
  }
  win_barrier();
  if (rfl(ctl[C::kCtlFlag / 4 + phase]) == 0) return;
  int ulo8 = 0x7fffffff, uhi8 = -0x7fffffff, uloy = 0x7fffffff, uhiy = -0x7fffffff;
#pragma unroll
  for (int w = 0; w < C::kWaves; ++w) {
    ulo8 = min(ulo8, ctl[w * 4 + 0]);
    uhi8 = max(uhi8, ctl[w * 4 + 1]);
    uloy = min(uloy, ctl[w * 4 + 2]);
    uhiy = max(uhiy, ctl[w * 4 + 3]);
  }
  const float wvx = __int_as_float(ctl[C::kCtlVel / 4 + 0]), wvy = __int_as_float(ctl[C::kCtlVel / 4 + 1]);
  // first and last texel the tile touches now (right / lower tap included), relative to the current origin
  const int bx0 = ulo8 >> 3, bx1 = (uhi8 >> 3) + 1, by0 = uloy, by1 = uhiy + 1;
  const int slack_x = max(C::kW - (bx1 - bx0 + 1), 0), slack_y = max(C::kH - (by1 - by0 + 1), 0);
  // texels kept on the low side: all the slack but two where the motion goes that way, two where it comes
  // from, half of it in calm air
  const int keep_x = wvx > 0.125f ? max(slack_x - 2, 0) : (wvx < -0.125f ? min(slack_x, 2) : slack_x / 2);
  const int keep_y = wvy > 0.125f ? max(slack_y - 2, 0) : (wvy < -0.125f ? min(slack_y, 2) : slack_y / 2);
  const int nox = rfl(min(max((W.ox + bx0 - keep_x) & ~3, 0), n - C::kW));  // 16-byte aligned rows of both planes
  const int noy = rfl(min(max(W.oy + by0 - keep_y, 0), m - C::kH));
  // (a tile parked at the image border keeps asking: the window it would get is the one it has)
  if (!force && nox == W.ox && noy == W.oy) return;
  win_count(W, 2);
  const int ddx8 = (nox - W.ox) * 8, ddy = noy - W.oy;
#pragma unroll
  for (int j = 0; j < kWinRows; ++j) {
    dx8[j] -= ddx8;
    dy[j] -= ddy;
  }
  W.ox = nox;
  W.oy = noy;
  // every wave is past the barrier: nobody reads the old window any more
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // opaque: the item addresses below are not loop invariants worth 40 registers
  constexpr int kRoundsUV = (C::kItemsUV + C::kThreads - 1) / C::kThreads, kRoundsP = (C::kItemsP + C::kThreads - 1) / C::kThreads;
  // (threads past the last item repeat it: the same bytes to the same place, and no exec-masked rounds)
  u32x4 buv[kRoundsUV], bp[kRoundsP];
  const unsigned org = static_cast<unsigned>(noy) * static_cast<unsigned>(n) + static_cast<unsigned>(nox);
#pragma unroll
  for (int k = 0; k < kRoundsUV; ++k) {
    const int item = min(tid + C::kThreads * k, C::kItemsUV - 1);
    const int row = item / (C::kW / 2), c = item - row * (C::kW / 2);
    buv[k] = __builtin_amdgcn_raw_buffer_load_b128(F.ruv, static_cast<int>((org + row * n + 2 * c) << 3), 0, 0);
  }
#pragma unroll
  for (int k = 0; k < kRoundsP; ++k) {
    const int item = min(tid + C::kThreads * k, C::kItemsP - 1);
    const int row = item / (C::kW / 4), c = item - row * (C::kW / 4);
    bp[k] = __builtin_amdgcn_raw_buffer_load_b128(F.rp, static_cast<int>((org + row * n + 4 * c) << 2), 0, 0);
  }
#pragma unroll
  for (int k = 0; k < kRoundsUV; ++k)
    *(lds_u32x4 *)(size_t)(W.uv + 16u * min(tid + C::kThreads * k, C::kItemsUV - 1)) = buv[k];
#pragma unroll
  for (int k = 0; k < kRoundsP; ++k)
    *(lds_u32x4 *)(size_t)(W.p + 16u * min(tid + C::kThreads * k, C::kItemsP - 1)) = bp[k];
  win_barrier();
}

// retreat() on the pre-scaled column offset (8 bytes per pixel): same t, same floor, same fraction
__device__ __forceinline__ void retreat8(int &P8, float &f, float w) {
  const float t = f - w;
  int k;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k) : "v"(t));
  P8 += k * 8;
  f = __builtin_amdgcn_fractf(t);
}

template <class C, bool GEN>
__global__ __launch_bounds__(C::kThreads, C::kWaves == 4 ? 3 : 4) void semilag_window(
    const float *__restrict__ precip, const float *__restrict__ vel, const float *__restrict__ vel_packed,
    float *__restrict__ out, double *__restrict__ disp, const float *__restrict__ scale, float first_scale, int m,
    int n, int T, int n_iter, int resume, float outval, int row0, int rows, int bmode, int tiles_x, int n_tiles,
    int tiles_per_xcd, unsigned long long *__restrict__ stats) {
  const int b = blockIdx.x;
  const int tile = (b % kNumXcd) * tiles_per_xcd + b / kNumXcd;
  if (tile >= n_tiles) return;  // the whole workgroup
  const int lane = threadIdx.x & (kTileX - 1);
  const int xt = (tile % tiles_x) * kTileX + lane;
  const int yt = row0 + (tile / tiles_x) * C::kTileY + static_cast<int>(threadIdx.x / kTileX) * kWinRows;
  const int x = min(xt, n - 1);
  const size_t plane = static_cast<size_t>(m) * n;
  Fields F;
  F.u0 = vel;
  F.v0 = vel + plane;
  F.p0 = precip;
  const int plane_bytes = static_cast<int>(plane * sizeof(float));
  F.ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel), 0, plane_bytes, 0x00020000);
  F.rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel + plane), 0, plane_bytes, 0x00020000);
  F.rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(precip), 0, plane_bytes, 0x00020000);
  F.ruv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(vel_packed), 0, 2 * plane_bytes, 0x00020000);
  F.rpp = F.ruv;  // (no row-pair field plane in this kernel)
  F.row_bytes = n * static_cast<int>(sizeof(float));
  F.coef = nullptr;
  F.cpad = 0;
  F.minval = 0.f;
  F.bmode = bmode;

  __shared__ __attribute__((aligned(16))) float win_uv[C::kW * C::kH * 2];
  __shared__ __attribute__((aligned(16))) float win_p[C::kW * C::kH];
  __shared__ __attribute__((aligned(16))) int win_ctl[C::kCtlWords];
  Window W;
  W.uv = static_cast<unsigned>(reinterpret_cast<size_t>((lds_void *)win_uv));
  W.p = static_cast<unsigned>(reinterpret_cast<size_t>((lds_void *)win_p));
  W.ctl = static_cast<unsigned>(reinterpret_cast<size_t>((lds_void *)win_ctl));
  W.ox = W.oy = 0;
  W.stats = stats;
  if (threadIdx.x < 3) win_ctl[C::kCtlFlag / 4 + threadIdx.x] = 0;

  int y[kWinRows], dx8[kWinRows], dy[kWinRows];
  float fx[kWinRows], fy[kWinRows], vix[kWinRows], viy[kWinRows], su[kWinRows], sv[kWinRows], sp[kWinRows];
  bool live[kWinRows];
  unsigned opix[kWinRows];
#pragma unroll
  for (int j = 0; j < kWinRows; ++j) {
    live[j] = xt < n && yt + j < row0 + rows;
    y[j] = min(yt + j, m - 1);
    opix[j] = static_cast<unsigned>(__mul24(y[j] - row0, n) + x) << 2;
    int px = x, py = y[j];
    fx[j] = fy[j] = sp[j] = 0.f;
    if (resume) {
      split_displacement(disp[static_cast<size_t>(y[j]) * n + x], px, fx[j]);
      split_displacement(disp[plane + static_cast<size_t>(y[j]) * n + x], py, fy[j]);
    }
    dx8[j] = px * 8;  // relative to the origin (0, 0) until the first window is placed
    dy[j] = py;
    vix[j] = viy[j] = 0.f;
  }
  const float move_scale = 2.5f * static_cast<float>(n_iter);  // lead step = n_iter sub-steps of two half increments, + 25 %
  if (!resume) {
    // first increment is NOT divided by n_iter (semilagrangian.py:202)
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      const unsigned pix = static_cast<unsigned>(__mul24(y[j], n) + x) << 2;
      vix[j] = ld(F.u0, pix) * first_scale;
      viy[j] = ld(F.v0, pix) * first_scale;
    }
  }
  __syncthreads();  // the flag words are cleared
  int phase = 0;
  win_update<C>(F, W, phase, true, dx8, dy, 0.5f * vix[0], 0.5f * viy[0], move_scale, m, n);
  phase = 1;
  if (resume) {
    win_sample<C, kVel, GEN>(F, W, dx8, dy, fx, fy, m, n, outval, su, sv, sp);
    const float s0 = scale[0];
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      vix[j] = su[j] * s0;
      viy[j] = sv[j] * s0;
    }
  }
  const float lostval = (bmode == kModeNearest || bmode == kModeGridConstant) ? __builtin_nanf("") : outval;
  // the increment is only ever used halved (midpoint rule): carry Vi / 2 (exact)
#pragma unroll
  for (int j = 0; j < kWinRows; ++j) {
    vix[j] *= 0.5f;
    viy[j] *= 0.5f;
  }

  for (int t = 0; t < T; ++t) {
    const float s = scale[t];
    const float half_s = 0.5f * s;
    for (int k = 0; k < n_iter; ++k) {
      int mx8[kWinRows], my[kWinRows];
      float gx[kWinRows], gy[kWinRows];
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        mx8[j] = dx8[j];
        my[j] = dy[j];
        gx[j] = fx[j];
        gy[j] = fy[j];
        retreat8(mx8[j], gx[j], vix[j]);  // midpoint rule (:213), vix = Vi / 2
        retreat(my[j], gy[j], viy[j]);
      }
      win_sample<C, kVel, GEN>(F, W, mx8, my, gx, gy, m, n, outval, su, sv, sp);
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        retreat8(dx8[j], fx[j], su[j] * s);
        retreat(dy[j], fy[j], sv[j] * s);
      }
      if (k == n_iter - 1) {
        win_sample<C, kVel | kPrecip, GEN>(F, W, dx8, dy, fx, fy, m, n, outval, su, sv, sp);
      } else {
        win_sample<C, kVel, GEN>(F, W, dx8, dy, fx, fy, m, n, outval, su, sv, sp);
      }
#pragma unroll
      for (int j = 0; j < kWinRows; ++j) {
        vix[j] = su[j] * half_s;
        viy[j] = sv[j] * half_s;
      }
    }
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      sp[j] = lost(fx[j], fy[j]) ? lostval : sp[j];
      if (live[j]) __builtin_nontemporal_store(sp[j], reinterpret_cast<float *>(reinterpret_cast<char *>(out) + opix[j]));
    }
    out += static_cast<size_t>(rows) * n;
    if (t + 1 < T) {
      win_update<C>(F, W, phase, false, dx8, dy, vix[0], viy[0], move_scale, m, n);
      phase = phase == 2 ? 0 : phase + 1;
    }
  }

  if (disp != nullptr) {
#pragma unroll
    for (int j = 0; j < kWinRows; ++j) {
      if (!live[j]) continue;
      disp[static_cast<size_t>(y[j]) * n + x] =
          static_cast<double>(W.ox + (dx8[j] >> 3) - x) + static_cast<double>(fx[j]);
      disp[plane + static_cast<size_t>(y[j]) * n + x] =
          static_cast<double>(W.oy + dy[j] - y[j]) + static_cast<double>(fy[j]);
    }
  }
}

bool semilag_window_eligible(const SemilagArgs &a) {
  return a.precip != nullptr && a.vel_packed != nullptr && a.order == 1 && a.n_iter >= 1 && a.n % 4 == 0 &&
         a.n >= Win8::kW && a.m >= Win8::kH && reinterpret_cast<uintptr_t>(a.vel_packed) % 16 == 0 &&
         reinterpret_cast<uintptr_t>(a.precip) % 16 == 0;
}

// debug counters of the window kernels (PYSTEPS_HIP_SL_STATS=1): printed after every launch, which then waits
static unsigned long long *g_win_stats = nullptr;

template <class C>
static hipError_t launch_window(const SemilagArgs &a, hipStream_t stream) {
  const int tiles_x = (a.n + kTileX - 1) / kTileX;
  const int tiles_y = (a.rows + C::kTileY - 1) / C::kTileY;
  const int n_tiles = tiles_x * tiles_y;
  const int tiles_per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
  const dim3 grid(tiles_per_xcd * kNumXcd), block(C::kThreads);
  static const bool want_stats = std::getenv("PYSTEPS_HIP_SL_STATS") != nullptr;
  if (want_stats) {
    if (g_win_stats == nullptr && hipMalloc(&g_win_stats, 4 * sizeof(unsigned long long)) != hipSuccess) g_win_stats = nullptr;
    if (g_win_stats != nullptr) (void)hipMemsetAsync(g_win_stats, 0, 4 * sizeof(unsigned long long), stream);
  }
  if (a.bmode != 0) {
    hipLaunchKernelGGL((semilag_window<C, true>), grid, block, 0, stream, a.precip, a.vel, a.vel_packed, a.out, a.disp,
                       a.scale, a.first_scale, a.m, a.n, a.T, a.n_iter, a.resume, a.outval, a.row0, a.rows, a.bmode, tiles_x,
                       n_tiles, tiles_per_xcd, g_win_stats);
  } else {
    hipLaunchKernelGGL((semilag_window<C, false>), grid, block, 0, stream, a.precip, a.vel, a.vel_packed, a.out, a.disp,
                       a.scale, a.first_scale, a.m, a.n, a.T, a.n_iter, a.resume, a.outval, a.row0, a.rows, a.bmode, tiles_x,
                       n_tiles, tiles_per_xcd, g_win_stats);
  }
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess && want_stats && g_win_stats != nullptr) {
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipMemcpyAsync(h, g_win_stats, sizeof(h), hipMemcpyDeviceToHost, stream) == hipSuccess &&
        hipStreamSynchronize(stream) == hipSuccess)
      std::fprintf(stderr, "semilag_window<%d waves>: %dx%d T=%d: wave-passes through the window %llu, through the gathers %llu, "
                   "window fills %llu (of %d workgroups x %d lead steps)\n", C::kWaves, a.m, a.n, a.T, h[0], h[1], h[2],
                   n_tiles, a.T);
  }
  return e;
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

// 0 (default): velocity gathered from a packed {u,v} plane and the field from a row-pair plane (dwordx4 gathers);
// 5: packed velocity only; 1: one plane per component with DPP column sharing (what short calls take anyway);
// 8: per-wave LDS staging of the packed planes; 9 / 10: workgroup window kept in LDS across lead steps.
static int g_semilag_variant = [] {
  const char *e = std::getenv("PYSTEPS_HIP_SL_VARIANT");
  return e ? std::atoi(e) : 0;
}();

void set_semilag_variant(int v) { g_semilag_variant = v; }

hipError_t launch_semilag(const SemilagArgs &a, hipStream_t stream) {
  // LDS staging needs 16-byte aligned rows (n % 4 == 0)
  const bool aligned = (a.n % 4 == 0) && (reinterpret_cast<uintptr_t>(a.vel) % 16 == 0) &&
                       (a.precip == nullptr || reinterpret_cast<uintptr_t>(a.precip) % 16 == 0);
  if (g_semilag_variant == 9 && semilag_window_eligible(a)) return launch_window<Win4>(a, stream);
  if (g_semilag_variant == 10 && semilag_window_eligible(a)) return launch_window<Win8>(a, stream);
  if (g_semilag_variant == 11 && semilag_window_eligible(a)) return launch_window<Win8Raw>(a, stream);
  if (a.vel_packed != nullptr && a.field_pairs != nullptr && a.order == 1) return launch_variant<1, kModePacked2>(a, stream);
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
  return (g_semilag_variant == 0 || g_semilag_variant == 5 || g_semilag_variant == 8 || g_semilag_variant >= 9) &&
         static_cast<uint64_t>(a.m) * static_cast<uint64_t>(a.n) < (1ull << 29) &&
         static_cast<long long>(a.T) * (a.n_iter > 0 ? a.n_iter : 1) >= 8;
}
// variant 0 also samples the field from a row-pair plane (one dwordx4 per sample); 5 = packed
// velocity only (two dwordx2 for the field), kept for comparison; variant 8 stages the plain field
// plane through LDS and needs no second copy of it
bool semilag_wants_field_pairs(const SemilagArgs &a) {
  return g_semilag_variant == 0 && semilag_wants_packed(a) && a.precip != nullptr &&
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
