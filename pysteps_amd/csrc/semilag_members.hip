// Member-batched, stateful semi-Lagrangian step for ensemble nowcasts (gfx950).
//
// Serves the calling convention of the generic nowcast loop,
// pysteps/nowcasts/utils.py:441-462 (worker1): for every ensemble member j
//     velocity_j = velocity + velocity_pert_gen[j](t)                     (:448-451)
//     precip_j, D_j = extrapolator(precip_j, velocity_j, [dt], displacement_prev=D_j,
//                                  return_displacement=True)               (:453-458)
// in ONE launch for all members, with every D_j resident in HBM between calls.
// The BPS motion perturbation (pysteps/noise/motion.py:146-180, generate_bps) is
//     velocity_j = V + a_j * V_par + b_j * V_perp,   V_par = V/|V|, V_perp = (-V_par_y, V_par_x)
// with two scalars per member (a_j = g_par(t) eps_par_j / vsf, b_j likewise), so the 48
// perturbed velocity fields are never materialised: the kernel samples V and the
// unit field V_par at the same taps (bilinear interpolation is linear) and combines
// them with the member's scalars held in SGPRs.
//
// One thread = one pixel of one member (grid.y = member); same trajectory
// arithmetic and resampling rules as semilag.hip (shared header), direct gathers
// through buffer descriptors; roofline = HBM, 48 B/pixel/step algorithmic at
// n_iter=1 (D read+write 32, precip in/out 8, velocity passes amortised by L2).
#include <algorithm>

#include "common.h"

#pragma clang fp contract(off)

#include "semilag_device.h"

namespace psh {
int g_members_variant = 2;  // members per thread of the packed kernel: 2 (default) or 1
void set_members_variant(int v) { g_members_variant = v; }
namespace {

using namespace sl;

struct Planes {
  __amdgpu_buffer_rsrc_t u, v, hu, hv, p;  // velocity, unit velocity, this member's precip
  __amdgpu_buffer_rsrc_t packed;           // {u,v,hu,hv} float4 (PERT) or {u,v} float2 per pixel, or unused
  const float *pu, *pv, *phu, *phv, *pp;   // the same as raw pointers (border path)
  int row_bytes;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 bld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(byte_off), soff, 0));
}
__device__ __forceinline__ f32x2 splat(float w) { return f32x2{w, w}; }

// blend() of semilag_device.h on two components at once (v_pk_mul_f32 / v_pk_fma_f32), same
// operation order per component: bit-identical to the one-plane-per-component path
__device__ __forceinline__ f32x2 blend2(const Weights &w, f32x2 a, f32x2 b, f32x2 c, f32x2 d) {
  f32x2 acc = a * w.w00;
  acc = __builtin_elementwise_fma(splat(w.w01), b, acc);
  acc = __builtin_elementwise_fma(splat(w.w10), c, acc);
  return __builtin_elementwise_fma(splat(w.w11), d, acc);
}

__device__ __forceinline__ float bld(__amdgpu_buffer_rsrc_t r, unsigned byte_off, int soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, static_cast<int>(byte_off), soff, 0));
}

__device__ __forceinline__ float tap4(__amdgpu_buffer_rsrc_t r, unsigned off, int rb, const Weights &w) {
  return blend(w, bld(r, off, 0), bld(r, off + 4u, 0), bld(r, off, rb), bld(r, off + 4u, rb));
}

__device__ __forceinline__ float tap4_clamped(const float *p, int X, int Y, const Weights &w, int m, int n) {
  const int x0 = min(max(X, 0), n - 1), x1 = min(max(X + 1, 0), n - 1);
  const int y0 = min(max(Y, 0), m - 1), y1 = min(max(Y + 1, 0), m - 1);
  const unsigned r0 = static_cast<unsigned>(__mul24(y0, n)), r1 = static_cast<unsigned>(__mul24(y1, n));
  return blend(w, ld(p, (r0 + x0) << 2), ld(p, (r0 + x1) << 2), ld(p, (r1 + x0) << 2), ld(p, (r1 + x1) << 2));
}

// velocity of this member at (X + fx, Y + fy), mode="nearest"; optionally the precip sample too
template <int ORDER, bool PERT, bool WITH_P, bool PACKED>
__device__ __forceinline__ void sample_member(const Planes &F, int X, int Y, float fx, float fy, int m,
                                              int n, float a, float b, float outval, float &su,
                                              float &sv, float &sp) {
  const Weights w = make_weights(fx, fy);
  float hu = 0.f, hv = 0.f;
  if (PACKED && wave_all_interior(X, Y, m, n)) {
    // A vector memory instruction costs the pipeline by its width class, not by its bytes
    // (tools/gather_probe.py: dword ~9.5 clk, dwordx4 ~17 clk per wave): with the velocity and the
    // unit velocity interleaved per pixel one dwordx4 delivers a whole tap - 4 loads per sampling
    // pass instead of 16 (perturbed) / 2 instead of 8 (unperturbed, {u,v} pairs: both columns of a
    // tap row in one load).
    const unsigned off = static_cast<unsigned>(__mul24(Y, n) + X) << 2;
    const int rb = F.row_bytes;
    f32x2 uv;
    if (PERT) {
      const unsigned o4 = off << 2;
      const f32x4 a0 = bld4(F.packed, o4, 0), a1 = bld4(F.packed, o4 + 16u, 0);
      const f32x4 b0 = bld4(F.packed, o4, 4 * rb), b1 = bld4(F.packed, o4 + 16u, 4 * rb);
      uv = blend2(w, a0.xy, a1.xy, b0.xy, b1.xy);
      const f32x2 h = blend2(w, a0.zw, a1.zw, b0.zw, b1.zw);
      hu = h.x;
      hv = h.y;
    } else {
      const unsigned o2 = off << 1;
      const f32x4 t = bld4(F.packed, o2, 0), bb = bld4(F.packed, o2, 2 * rb);
      uv = blend2(w, t.xy, t.zw, bb.xy, bb.zw);
    }
    su = uv.x;
    sv = uv.y;
    if (WITH_P) {
      if (ORDER == 1) {
        const f32x2 pt = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(F.p, static_cast<int>(off), 0, 0));
        const f32x2 pb = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(F.p, static_cast<int>(off), rb, 0));
        sp = blend(w, pt.x, pt.y, pb.x, pb.y);
      } else {
        const int xi = X + (fx >= 0.5f ? 1 : 0), yi = Y + (fy >= 0.5f ? 1 : 0);
        sp = bld(F.p, static_cast<unsigned>(__mul24(yi, n) + xi) << 2, 0);
      }
    }
    asm volatile("" ::: "memory");
  } else if (wave_all_interior(X, Y, m, n)) {
    const unsigned off = static_cast<unsigned>(__mul24(Y, n) + X) << 2;
    su = tap4(F.u, off, F.row_bytes, w);
    sv = tap4(F.v, off, F.row_bytes, w);
    if (PERT) {
      hu = tap4(F.hu, off, F.row_bytes, w);
      hv = tap4(F.hv, off, F.row_bytes, w);
    }
    if (WITH_P) {
      if (ORDER == 1) {
        sp = tap4(F.p, off, F.row_bytes, w);
      } else {
        const int xi = X + (fx >= 0.5f ? 1 : 0), yi = Y + (fy >= 0.5f ? 1 : 0);
        sp = bld(F.p, static_cast<unsigned>(__mul24(yi, n) + xi) << 2, 0);
      }
    }
    asm volatile("" ::: "memory");  // keep the two paths from being merged (see semilag.hip)
  } else {
    su = tap4_clamped(F.pu, X, Y, w, m, n);
    sv = tap4_clamped(F.pv, X, Y, w, m, n);
    if (PERT) {
      hu = tap4_clamped(F.phu, X, Y, w, m, n);
      hv = tap4_clamped(F.phv, X, Y, w, m, n);
    }
    if (WITH_P) sp = sample_precip_border<ORDER>(F.pp, X, Y, fx, fy, m, n, outval);
  }
  if (PERT) {
    // V + a * V_par + b * V_perp with V_perp = (-V_par_y, V_par_x)
    su = su + (a * hu - b * hv);
    sv = sv + (a * hv + b * hu);
  }
}

// the advected field alone at (X + fx, Y + fy): the sample of the LAST sub-step of a call, whose
// velocity sample nobody would read (the next call rebuilds its increment from the stored position)
template <int ORDER>
__device__ __forceinline__ float sample_field_only(const Planes &F, int X, int Y, float fx, float fy, int m, int n,
                                                   float outval) {
  float sp;
  if (wave_all_interior(X, Y, m, n)) {
    const unsigned off = static_cast<unsigned>(__mul24(Y, n) + X) << 2;
    if (ORDER == 1) {
      const Weights w = make_weights(fx, fy);
      const f32x2 pt = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(F.p, static_cast<int>(off), 0, 0));
      const f32x2 pb = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(F.p, static_cast<int>(off), F.row_bytes, 0));
      sp = blend(w, pt.x, pt.y, pb.x, pb.y);
    } else {
      const int xi = X + (fx >= 0.5f ? 1 : 0), yi = Y + (fy >= 0.5f ? 1 : 0);
      sp = bld(F.p, static_cast<unsigned>(__mul24(yi, n) + xi) << 2, 0);
    }
    asm volatile("" ::: "memory");
  } else {
    sp = sample_precip_border<ORDER>(F.pp, X, Y, fx, fy, m, n, outval);
  }
  return sp;
}

// State of a trajectory between calls.  COMPACT = the kernel's own representation, one 16-byte
// record per pixel and member: integer pixel offsets (P - x, P - y) and the two fractions as
// float32 - half the bytes of the float64 displacement pair of the reference, read and written
// with one dwordx4 access each, and no float64 conversions in the kernel.
template <int ORDER, bool PERT, bool HAS_PRECIP, bool COMPACT, bool PACKED>
__global__ __launch_bounds__(256) void semilag_members(
    const float *__restrict__ precip, const float *__restrict__ vel, const float *__restrict__ vhat,
    const float *__restrict__ packed,
    const float *__restrict__ pert_ab, float *__restrict__ out, void *__restrict__ state,
    const float *__restrict__ scale, float first_scale, int m, int n, int T, int n_iter, int resume,
    float outval, int tiles_x, int n_tiles, int tiles_per_xcd, int n_members) {
  // XCD-contiguous bands of tiles, and within an XCD the members of a tile back to back (member is the
  // fastest index of the XCD's block sequence): the members sample the SAME velocity planes around the
  // same pixels at the same time, so one member's gathers leave the lines in the XCD's L2 for the others
  // - 1.6 GB of the 5.5 GB a 6-member launch fetched from HBM were re-fetches of the packed plane
  // (profiles/r03/a_members_pmc_traffic.json)
  const int blk = blockIdx.x;
  const int seq = blk / kNumXcd;
  const int tile = (blk % kNumXcd) * tiles_per_xcd + seq / n_members;
  if (tile >= n_tiles) return;
  const int member = seq % n_members;
  const int x = (tile % tiles_x) * 64 + (threadIdx.x & 63);
  const int y = (tile / tiles_x) * 4 + (threadIdx.x >> 6);
  const bool live = x < n && y < m;
  const int xc = min(x, n - 1), yc = min(y, m - 1);
  const size_t plane = static_cast<size_t>(m) * n;
  const int plane_bytes = static_cast<int>(plane * sizeof(float));

  Planes F;
  F.pu = vel;
  F.pv = vel + plane;
  F.phu = PERT ? vhat : vel;
  F.phv = PERT ? vhat + plane : vel;
  F.pp = HAS_PRECIP ? precip + static_cast<size_t>(member) * plane : vel;
  F.u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F.pu), 0, plane_bytes, 0x00020000);
  F.v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F.pv), 0, plane_bytes, 0x00020000);
  F.hu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F.phu), 0, plane_bytes, 0x00020000);
  F.hv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F.phv), 0, plane_bytes, 0x00020000);
  F.p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F.pp), 0, plane_bytes, 0x00020000);
  F.packed = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(PACKED ? packed : vel), 0,
                                               (PERT ? 4 : 2) * plane_bytes, 0x00020000);
  F.row_bytes = n * static_cast<int>(sizeof(float));
  const float a = PERT ? pert_ab[2 * member] : 0.f, b = PERT ? pert_ab[2 * member + 1] : 0.f;

  double *dplane = static_cast<double *>(state) + static_cast<size_t>(member) * 2 * plane;
  const size_t pix = static_cast<size_t>(yc) * n + xc;
  uint4 *record = static_cast<uint4 *>(state) + static_cast<size_t>(member) * plane + pix;
  int px = xc, py = yc;
  float fx = 0.f, fy = 0.f, vix, viy, su, sv, sp = 0.f;
  if (resume) {
    if (COMPACT) {
      const uint4 r = *record;
      px += static_cast<int>(r.x);
      py += static_cast<int>(r.y);
      fx = __uint_as_float(r.z);
      fy = __uint_as_float(r.w);
    } else {
      split_displacement(dplane[pix], px, fx);
      split_displacement(dplane[plane + pix], py, fy);
    }
    sample_member<ORDER, PERT, false, PACKED>(F, px, py, fx, fy, m, n, a, b, outval, su, sv, sp);
    vix = su * scale[0];
    viy = sv * scale[0];
  } else {
    sample_member<ORDER, PERT, false, PACKED>(F, px, py, 0.f, 0.f, m, n, a, b, outval, su, sv, sp);
    vix = su * first_scale;  // the very first increment is not divided by n_iter (:202)
    viy = sv * first_scale;
  }
  float *optr = out + (static_cast<size_t>(member) * T) * plane + pix;
  for (int t = 0; t < T; ++t) {
    const float s = scale[t];
    if (n_iter > 0) {
      for (int k = 0; k < n_iter; ++k) {
        int mx = px, my = py;
        float gx = fx, gy = fy;
        retreat(mx, gx, 0.5f * vix);
        retreat(my, gy, 0.5f * viy);
        sample_member<ORDER, PERT, false, PACKED>(F, mx, my, gx, gy, m, n, a, b, outval, su, sv, sp);
        retreat(px, fx, su * s);
        retreat(py, fy, sv * s);
        if (t == T - 1 && k == n_iter - 1) {
          // last sub-step of the call: the increment it would prepare is rebuilt by the next call from
          // the stored position (resume), so only the field is sampled - 4 of the 16 gathers of a
          // single-step call were these dead velocity taps
          if (HAS_PRECIP) sp = sample_field_only<ORDER>(F, px, py, fx, fy, m, n, outval);
        } else {
          if (HAS_PRECIP && k == n_iter - 1) {
            sample_member<ORDER, PERT, true, PACKED>(F, px, py, fx, fy, m, n, a, b, outval, su, sv, sp);
          } else {
            sample_member<ORDER, PERT, false, PACKED>(F, px, py, fx, fy, m, n, a, b, outval, su, sv, sp);
          }
          vix = su * s;
          viy = sv * s;
        }
      }
    } else {
      if (t > 0 || resume) {
        sample_member<ORDER, PERT, false, PACKED>(F, px, py, fx, fy, m, n, a, b, outval, su, sv, sp);
        vix = su * s;
        viy = sv * s;
      }
      retreat(px, fx, vix);
      retreat(py, fy, viy);
      if (HAS_PRECIP) {
        float du, dv;
        sample_member<ORDER, false, true, false>(F, px, py, fx, fy, m, n, 0.f, 0.f, outval, du, dv, sp);
      }
    }
    if (HAS_PRECIP) {
      // trajectories that met a non-finite velocity sample cval, like map_coordinates at a NaN coordinate
      if (live) *optr = lost(fx, fy) ? outval : sp;
      optr += plane;
    }
  }
  if (live) {
    if (COMPACT) {
      *record = make_uint4(static_cast<unsigned>(px - xc), static_cast<unsigned>(py - yc), __float_as_uint(fx),
                           __float_as_uint(fy));
    } else {
      dplane[pix] = static_cast<double>(px) - static_cast<double>(xc) + static_cast<double>(fx);
      dplane[plane + pix] = static_cast<double>(py) - static_cast<double>(yc) + static_cast<double>(fy);
    }
  }
}

// ---- two members per thread ----------------------------------------------------------------------
// A single-step call is a chain of four dependent memory round trips per trajectory (record ->
// increment rebuild -> midpoint -> field): with one trajectory per thread and 8 waves per SIMD the
// kernel ran at the latency of that chain (1.16 ms for 6 members at 4096^2, TA 60 % busy, HBM at 3.7
// of 8 TB/s).  Here a thread carries the same pixel of TWO members: both chains' gathers are issued
// before either is waited for, and the two members' taps fall into the same cache lines (their
// trajectories differ by the perturbation only).  Packed planes, compact records, n_iter >= 1.
struct PairTaps {
  f32x4 a0, a1, b0, b1;
  f32x2 pt, pb;
  float p0;
};

template <int ORDER, bool PERT, bool WITH_P>
__device__ __forceinline__ void issue_taps(const Planes &F, __amdgpu_buffer_rsrc_t prec, int X, int Y, float fx, float fy,
                                           int n, PairTaps &t) {
  const unsigned off = static_cast<unsigned>(__mul24(Y, n) + X) << 2;
  const int rb = F.row_bytes;
  if (PERT) {
    const unsigned o4 = off << 2;
    t.a0 = bld4(F.packed, o4, 0);
    t.a1 = bld4(F.packed, o4 + 16u, 0);
    t.b0 = bld4(F.packed, o4, 4 * rb);
    t.b1 = bld4(F.packed, o4 + 16u, 4 * rb);
  } else {
    const unsigned o2 = off << 1;
    t.a0 = bld4(F.packed, o2, 0);
    t.b0 = bld4(F.packed, o2, 2 * rb);
  }
  if (WITH_P) {
    if (ORDER == 1) {
      t.pt = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(prec, static_cast<int>(off), 0, 0));
      t.pb = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(prec, static_cast<int>(off), rb, 0));
    } else {
      const int xi = X + (fx >= 0.5f ? 1 : 0), yi = Y + (fy >= 0.5f ? 1 : 0);
      t.p0 = bld(prec, static_cast<unsigned>(__mul24(yi, n) + xi) << 2, 0);
    }
  }
}

template <int ORDER, bool PERT, bool WITH_P>
__device__ __forceinline__ void finish_taps(const PairTaps &t, float fx, float fy, float a, float b, float &su, float &sv,
                                            float &sp) {
  const Weights w = make_weights(fx, fy);
  f32x2 uv;
  if (PERT) {
    uv = blend2(w, t.a0.xy, t.a1.xy, t.b0.xy, t.b1.xy);
    const f32x2 h = blend2(w, t.a0.zw, t.a1.zw, t.b0.zw, t.b1.zw);
    su = uv.x + (a * h.x - b * h.y);
    sv = uv.y + (a * h.y + b * h.x);
  } else {
    uv = blend2(w, t.a0.xy, t.a0.zw, t.b0.xy, t.b0.zw);
    su = uv.x;
    sv = uv.y;
  }
  if (WITH_P) sp = ORDER == 1 ? blend(w, t.pt.x, t.pt.y, t.pb.x, t.pb.y) : t.p0;
}

struct Traj {
  int px, py;
  float fx, fy, vix, viy, su, sv, sp;
};

// velocity (and optionally the field) of both members at their positions (X[i] + gx[i], Y[i] + gy[i])
template <int ORDER, bool PERT, bool WITH_P>
__device__ __forceinline__ void sample_two(const Planes (&F)[2], const int (&X)[2], const int (&Y)[2], const float (&gx)[2],
                                           const float (&gy)[2], int m, int n, const float (&a)[2], const float (&b)[2],
                                           float outval, Traj (&tr)[2]) {
  if (wave_all_interior(X[0], Y[0], m, n) && wave_all_interior(X[1], Y[1], m, n)) {
    PairTaps t0, t1;
    issue_taps<ORDER, PERT, WITH_P>(F[0], F[0].p, X[0], Y[0], gx[0], gy[0], n, t0);
    issue_taps<ORDER, PERT, WITH_P>(F[1], F[1].p, X[1], Y[1], gx[1], gy[1], n, t1);
    finish_taps<ORDER, PERT, WITH_P>(t0, gx[0], gy[0], a[0], b[0], tr[0].su, tr[0].sv, tr[0].sp);
    finish_taps<ORDER, PERT, WITH_P>(t1, gx[1], gy[1], a[1], b[1], tr[1].su, tr[1].sv, tr[1].sp);
    asm volatile("" ::: "memory");
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
      sample_member<ORDER, PERT, WITH_P, true>(F[i], X[i], Y[i], gx[i], gy[i], m, n, a[i], b[i], outval, tr[i].su, tr[i].sv,
                                               tr[i].sp);
  }
}

template <int ORDER, bool PERT, bool HAS_PRECIP>
__global__ __launch_bounds__(256) void semilag_members_pair(
    const float *__restrict__ precip, const float *__restrict__ vel, const float *__restrict__ vhat,
    const float *__restrict__ packed, const float *__restrict__ pert_ab, float *__restrict__ out, void *__restrict__ state,
    const float *__restrict__ scale, float first_scale, int m, int n, int T, int n_iter, int resume, float outval,
    int tiles_x, int n_tiles, int tiles_per_xcd, int n_members) {
  const int groups = (n_members + 1) >> 1;
  const int blk = blockIdx.x;
  const int seq = blk / kNumXcd;
  const int tile = (blk % kNumXcd) * tiles_per_xcd + seq / groups;  // XCD-contiguous bands, member pairs back to back
  if (tile >= n_tiles) return;
  const int first = (seq % groups) * 2;
  const int member[2] = {first, min(first + 1, n_members - 1)};
  const bool second = first + 1 < n_members;  // odd member count: the last thread column carries one member twice
  const int x = (tile % tiles_x) * 64 + (threadIdx.x & 63);
  const int y = (tile / tiles_x) * 4 + (threadIdx.x >> 6);
  const bool live = x < n && y < m;
  const int xc = min(x, n - 1), yc = min(y, m - 1);
  const size_t plane = static_cast<size_t>(m) * n;
  const int plane_bytes = static_cast<int>(plane * sizeof(float));
  const size_t pix = static_cast<size_t>(yc) * n + xc;

  Planes F[2];
  float a[2], b[2];
  uint4 *record[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    F[i].pu = vel;
    F[i].pv = vel + plane;
    F[i].phu = PERT ? vhat : vel;
    F[i].phv = PERT ? vhat + plane : vel;
    F[i].pp = HAS_PRECIP ? precip + static_cast<size_t>(member[i]) * plane : vel;
    F[i].u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F[i].pu), 0, plane_bytes, 0x00020000);
    F[i].v = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F[i].pv), 0, plane_bytes, 0x00020000);
    F[i].hu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F[i].phu), 0, plane_bytes, 0x00020000);
    F[i].hv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F[i].phv), 0, plane_bytes, 0x00020000);
    F[i].p = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(F[i].pp), 0, plane_bytes, 0x00020000);
    F[i].packed = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(packed), 0, (PERT ? 4 : 2) * plane_bytes, 0x00020000);
    F[i].row_bytes = n * static_cast<int>(sizeof(float));
    a[i] = PERT ? pert_ab[2 * member[i]] : 0.f;
    b[i] = PERT ? pert_ab[2 * member[i] + 1] : 0.f;
    record[i] = static_cast<uint4 *>(state) + static_cast<size_t>(member[i]) * plane + pix;
  }

  Traj tr[2];
  int X[2], Y[2];
  float gx[2], gy[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    tr[i].px = xc;
    tr[i].py = yc;
    tr[i].fx = tr[i].fy = 0.f;
    tr[i].sp = 0.f;
  }
  if (resume) {
    const uint4 r0 = *record[0], r1 = *record[1];
    tr[0].px += static_cast<int>(r0.x);
    tr[0].py += static_cast<int>(r0.y);
    tr[0].fx = __uint_as_float(r0.z);
    tr[0].fy = __uint_as_float(r0.w);
    tr[1].px += static_cast<int>(r1.x);
    tr[1].py += static_cast<int>(r1.y);
    tr[1].fx = __uint_as_float(r1.z);
    tr[1].fy = __uint_as_float(r1.w);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    X[i] = tr[i].px;
    Y[i] = tr[i].py;
    gx[i] = tr[i].fx;
    gy[i] = tr[i].fy;
  }
  sample_two<ORDER, PERT, false>(F, X, Y, gx, gy, m, n, a, b, outval, tr);
  {
    const float s0 = resume ? scale[0] : first_scale;  // the very first increment is not divided by n_iter (:202)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      tr[i].vix = tr[i].su * s0;
      tr[i].viy = tr[i].sv * s0;
    }
  }
  float *optr[2] = {out + (static_cast<size_t>(member[0]) * T) * plane + pix, out + (static_cast<size_t>(member[1]) * T) * plane + pix};
  for (int t = 0; t < T; ++t) {
    const float s = scale[t];
    for (int k = 0; k < n_iter; ++k) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        X[i] = tr[i].px;
        Y[i] = tr[i].py;
        gx[i] = tr[i].fx;
        gy[i] = tr[i].fy;
        retreat(X[i], gx[i], 0.5f * tr[i].vix);
        retreat(Y[i], gy[i], 0.5f * tr[i].viy);
      }
      sample_two<ORDER, PERT, false>(F, X, Y, gx, gy, m, n, a, b, outval, tr);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        retreat(tr[i].px, tr[i].fx, tr[i].su * s);
        retreat(tr[i].py, tr[i].fy, tr[i].sv * s);
        X[i] = tr[i].px;
        Y[i] = tr[i].py;
        gx[i] = tr[i].fx;
        gy[i] = tr[i].fy;
      }
      if (t == T - 1 && k == n_iter - 1) {
        // last sub-step of the call: only the field (the next call rebuilds the increment, see above)
        if (HAS_PRECIP) {
          if (wave_all_interior(X[0], Y[0], m, n) && wave_all_interior(X[1], Y[1], m, n)) {
            float v[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const unsigned off = static_cast<unsigned>(__mul24(Y[i], n) + X[i]) << 2;
              if (ORDER == 1) {
                const f32x2 pt = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(F[i].p, static_cast<int>(off), 0, 0));
                const f32x2 pb = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(F[i].p, static_cast<int>(off), F[i].row_bytes, 0));
                v[i] = blend(make_weights(gx[i], gy[i]), pt.x, pt.y, pb.x, pb.y);
              } else {
                const int xi = X[i] + (gx[i] >= 0.5f ? 1 : 0), yi = Y[i] + (gy[i] >= 0.5f ? 1 : 0);
                v[i] = bld(F[i].p, static_cast<unsigned>(__mul24(yi, n) + xi) << 2, 0);
              }
            }
            tr[0].sp = v[0];
            tr[1].sp = v[1];
            asm volatile("" ::: "memory");
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) tr[i].sp = sample_precip_border<ORDER>(F[i].pp, X[i], Y[i], gx[i], gy[i], m, n, outval);
          }
        }
      } else {
        if (HAS_PRECIP && k == n_iter - 1) {
          sample_two<ORDER, PERT, true>(F, X, Y, gx, gy, m, n, a, b, outval, tr);
        } else {
          sample_two<ORDER, PERT, false>(F, X, Y, gx, gy, m, n, a, b, outval, tr);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          tr[i].vix = tr[i].su * s;
          tr[i].viy = tr[i].sv * s;
        }
      }
    }
    if (HAS_PRECIP) {
      if (live) {
        *optr[0] = lost(tr[0].fx, tr[0].fy) ? outval : tr[0].sp;
        if (second) *optr[1] = lost(tr[1].fx, tr[1].fy) ? outval : tr[1].sp;
      }
      optr[0] += plane;
      optr[1] += plane;
    }
  }
  if (live) {
    *record[0] = make_uint4(static_cast<unsigned>(tr[0].px - xc), static_cast<unsigned>(tr[0].py - yc), __float_as_uint(tr[0].fx),
                            __float_as_uint(tr[0].fy));
    if (second)
      *record[1] = make_uint4(static_cast<unsigned>(tr[1].px - xc), static_cast<unsigned>(tr[1].py - yc),
                              __float_as_uint(tr[1].fx), __float_as_uint(tr[1].fy));
  }
}

// compact trajectory records <-> float64 displacement (B,2,m,n)
__global__ __launch_bounds__(256) void members_state_to_disp(const uint4 *__restrict__ state, size_t plane,
                                                             double *__restrict__ disp) {
  const size_t member = blockIdx.y;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < plane; i += stride) {
    const uint4 r = state[member * plane + i];
    disp[(2 * member) * plane + i] = static_cast<double>(static_cast<int>(r.x)) + static_cast<double>(__uint_as_float(r.z));
    disp[(2 * member + 1) * plane + i] = static_cast<double>(static_cast<int>(r.y)) + static_cast<double>(__uint_as_float(r.w));
  }
}

__global__ __launch_bounds__(256) void members_disp_to_state(const double *__restrict__ disp, size_t plane,
                                                             uint4 *__restrict__ state) {
  const size_t member = blockIdx.y;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < plane; i += stride) {
    int ox = 0, oy = 0;
    float fx, fy;
    split_displacement(disp[(2 * member) * plane + i], ox, fx);
    split_displacement(disp[(2 * member + 1) * plane + i], oy, fy);
    state[member * plane + i] =
        make_uint4(static_cast<unsigned>(ox), static_cast<unsigned>(oy), __float_as_uint(fx), __float_as_uint(fy));
  }
}

// V / |V| with zeros where |V| <= 1e-12 (noise/motion.py:127-131)
__global__ __launch_bounds__(256) void velocity_unit(const float *__restrict__ vel, size_t plane,
                                                     float *__restrict__ vhat) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < plane; i += stride) {
    const float u = vel[i], v = vel[plane + i];
    const float nrm = sqrtf(u * u + v * v);
    const bool ok = nrm > 1e-12f;
    vhat[i] = ok ? u / nrm : 0.f;
    vhat[plane + i] = ok ? v / nrm : 0.f;
  }
}

// {u,v,hu,hv} (vhat given) or {u,v} per pixel for the dwordx4 gathers of the packed kernels
__global__ __launch_bounds__(256) void members_pack(const float *__restrict__ vel, const float *__restrict__ vhat,
                                                    size_t plane, float *__restrict__ out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < plane; i += stride) {
    if (vhat) {
      reinterpret_cast<f32x4 *>(out)[i] = f32x4{vel[i], vel[plane + i], vhat[i], vhat[plane + i]};
    } else {
      reinterpret_cast<f32x2 *>(out)[i] = f32x2{vel[i], vel[plane + i]};
    }
  }
}

}  // namespace
}  // namespace psh

extern "C" int psh_members_pack_dev(const float *velocity_dev, const float *vhat_dev, int m, int n, float *packed_dev) {
  PSH_REQUIRE_INIT();
  if (m <= 0 || n <= 0 || !velocity_dev || !packed_dev) return psh::fail(PSH_EINVAL, "members_pack: invalid argument");
  if (static_cast<uint64_t>(m) * n >= (1ull << 28))
    return psh::fail(PSH_EUNSUPPORTED, "members_pack: m*n must be < 2^28 pixels (32-bit byte offsets)");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::members_pack, dim3(4096), dim3(256), 0, c.stream, velocity_dev, vhat_dev,
                     static_cast<size_t>(m) * n, packed_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_velocity_unit_dev(const float *velocity_dev, int m, int n, float *vhat_dev) {
  PSH_REQUIRE_INIT();
  if (m <= 0 || n <= 0 || !velocity_dev || !vhat_dev)
    return psh::fail(PSH_EINVAL, "velocity_unit: invalid argument");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * n;
  hipLaunchKernelGGL(psh::velocity_unit, dim3(2048), dim3(256), 0, c.stream, velocity_dev, plane, vhat_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

static int members_step(const float *precip_dev, const float *velocity_dev, const float *vhat_dev,
                        const float *packed_dev, const double *pert_par_host, const double *pert_perp_host, int n_members, int m,
                        int n, const double *steps_host, int T, int n_iter, int interp_order, float outval,
                        void *disp_dev, bool compact, int resume, float *out_dev) {
  PSH_REQUIRE_INIT();
  if (n_members <= 0 || n_members > 65535)
    return psh::fail(PSH_EINVAL, "semilag_members: member count %d out of range", n_members);
  if (m <= 0 || n <= 0 || static_cast<uint64_t>(m) * n >= (1ull << 29))
    return psh::fail(PSH_EINVAL, "semilag_members: invalid shape (%d,%d)", m, n);
  if (packed_dev && static_cast<uint64_t>(m) * n >= (1ull << 28))
    return psh::fail(PSH_EUNSUPPORTED, "semilag_members: packed planes need m*n < 2^28 pixels");
  if (T <= 0 || T > 1024) return psh::fail(PSH_EINVAL, "semilag_members: T must be in 1..1024");
  if (n_iter < 0) return psh::fail(PSH_EINVAL, "semilag_members: n_iter must be >= 0");
  if (interp_order != 0 && interp_order != 1)
    return psh::fail(PSH_EUNSUPPORTED, "semilag_members: interp_order %d not implemented", interp_order);
  if (!velocity_dev || !steps_host || !disp_dev)
    return psh::fail(PSH_EINVAL, "semilag_members: NULL velocity/steps/displacement");
  if (precip_dev && !out_dev) return psh::fail(PSH_EINVAL, "semilag_members: precip given but out is NULL");
  const bool pert = vhat_dev != nullptr;
  if (pert && (!pert_par_host || !pert_perp_host))
    return psh::fail(PSH_EINVAL, "semilag_members: unit velocity given without member scalars");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  // per-call constants: T scale factors + 2 scalars per member, through the pinned slot ring
  // (asynchronous: back-to-back member steps queue up without a host round trip)
  const size_t n_const = static_cast<size_t>(T) + 2 * static_cast<size_t>(n_members);
  if (n_const > psh::kConstSlotFloats)
    return psh::fail(PSH_EUNSUPPORTED, "semilag_members: T + 2 * members must be <= %zu", psh::kConstSlotFloats);
  float *h = nullptr;
  const float *d_const = nullptr;
  if (int rc = psh::const_slot(&h, &d_const)) return rc;
  const double sub = n_iter > 1 ? static_cast<double>(n_iter) : 1.0;
  for (int t = 0; t < T; ++t) h[t] = static_cast<float>(steps_host[t] / sub);
  for (int j = 0; j < n_members; ++j) {
    h[T + 2 * j] = pert ? static_cast<float>(pert_par_host[j]) : 0.f;
    h[T + 2 * j + 1] = pert ? static_cast<float>(pert_perp_host[j]) : 0.f;
  }
  PSH_HIP(hipMemcpyAsync(const_cast<float *>(d_const), h, n_const * sizeof(float), hipMemcpyHostToDevice, c.stream));
  const int tiles_x = (n + 63) / 64, tiles_y = (m + 3) / 4;
  const int n_tiles = tiles_x * tiles_y;
  const int tiles_per_xcd = (n_tiles + psh::kNumXcd - 1) / psh::kNumXcd;
  const dim3 grid(static_cast<unsigned>(tiles_per_xcd) * psh::kNumXcd * n_members), block(256);
  const float first_scale = static_cast<float>(steps_host[0]);
#define PSH_MEMBERS_P(ORDER, PERT, HASP, COMPACT, PACKED)                                                  \
  hipLaunchKernelGGL((psh::semilag_members<ORDER, PERT, HASP, COMPACT, PACKED>), grid, block, 0, c.stream, \
                     precip_dev, velocity_dev, vhat_dev, packed_dev, d_const + T, out_dev, disp_dev, d_const,  \
                     first_scale, m, n, T, n_iter, resume, outval, tiles_x, n_tiles, tiles_per_xcd, n_members)
#define PSH_MEMBERS_C(ORDER, PERT, HASP, COMPACT)         \
  do {                                                    \
    if (packed_dev) {                                     \
      PSH_MEMBERS_P(ORDER, PERT, HASP, COMPACT, true);    \
    } else {                                              \
      PSH_MEMBERS_P(ORDER, PERT, HASP, COMPACT, false);   \
    }                                                     \
  } while (0)
#define PSH_MEMBERS(ORDER, PERT, HASP)                  \
  do {                                                  \
    if (compact) {                                      \
      PSH_MEMBERS_C(ORDER, PERT, HASP, true);           \
    } else {                                            \
      PSH_MEMBERS_C(ORDER, PERT, HASP, false);          \
    }                                                   \
  } while (0)
  // packed planes + compact state (what EnsembleAdvector runs), n_iter >= 1: two members per thread
  if (packed_dev && compact && n_iter > 0 && n_members > 1 && psh::g_members_variant == 2) {
    const int groups = (n_members + 1) / 2;
    const dim3 grid2(static_cast<unsigned>(tiles_per_xcd) * psh::kNumXcd * groups);
#define PSH_PAIR(ORDER, PERT, HASP)                                                                        \
  hipLaunchKernelGGL((psh::semilag_members_pair<ORDER, PERT, HASP>), grid2, block, 0, c.stream, precip_dev, \
                     velocity_dev, vhat_dev, packed_dev, d_const + T, out_dev, disp_dev, d_const, first_scale, \
                     m, n, T, n_iter, resume, outval, tiles_x, n_tiles, tiles_per_xcd, n_members)
    if (!precip_dev) {
      if (pert) PSH_PAIR(1, true, false); else PSH_PAIR(1, false, false);
    } else if (interp_order == 0) {
      if (pert) PSH_PAIR(0, true, true); else PSH_PAIR(0, false, true);
    } else {
      if (pert) PSH_PAIR(1, true, true); else PSH_PAIR(1, false, true);
    }
#undef PSH_PAIR
  } else if (!precip_dev) {
    if (pert) PSH_MEMBERS(1, true, false); else PSH_MEMBERS(1, false, false);
  } else if (interp_order == 0) {
    if (pert) PSH_MEMBERS(0, true, true); else PSH_MEMBERS(0, false, true);
  } else {
    if (pert) PSH_MEMBERS(1, true, true); else PSH_MEMBERS(1, false, true);
  }
#undef PSH_MEMBERS_P
#undef PSH_MEMBERS_C
#undef PSH_MEMBERS
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return psh::fail(PSH_EHIP, "semilag_members launch failed: %s", hipGetErrorString(e));
  return PSH_OK;
}

extern "C" int psh_semilag_members_dev(const float *precip_dev, const float *velocity_dev,
                                       const float *vhat_dev, const double *pert_par_host,
                                       const double *pert_perp_host, int n_members, int m, int n,
                                       const double *steps_host, int T, int n_iter, int interp_order,
                                       float outval, double *disp_dev, int resume, float *out_dev) {
  return members_step(precip_dev, velocity_dev, vhat_dev, nullptr, pert_par_host, pert_perp_host, n_members, m, n,
                      steps_host, T, n_iter, interp_order, outval, disp_dev, false, resume, out_dev);
}

extern "C" int psh_semilag_members_state_dev(const float *precip_dev, const float *velocity_dev,
                                             const float *vhat_dev, const double *pert_par_host,
                                             const double *pert_perp_host, int n_members, int m, int n,
                                             const double *steps_host, int T, int n_iter, int interp_order,
                                             float outval, void *state_dev, int resume, float *out_dev) {
  return members_step(precip_dev, velocity_dev, vhat_dev, nullptr, pert_par_host, pert_perp_host, n_members, m, n,
                      steps_host, T, n_iter, interp_order, outval, state_dev, true, resume, out_dev);
}

extern "C" int psh_semilag_members_packed_dev(const float *precip_dev, const float *velocity_dev,
                                              const float *vhat_dev, const float *packed_dev,
                                              const double *pert_par_host, const double *pert_perp_host,
                                              int n_members, int m, int n, const double *steps_host, int T,
                                              int n_iter, int interp_order, float outval, void *state_dev,
                                              int resume, float *out_dev) {
  if (!packed_dev) return psh::fail(PSH_EINVAL, "semilag_members_packed: NULL packed plane");
  return members_step(precip_dev, velocity_dev, vhat_dev, packed_dev, pert_par_host, pert_perp_host, n_members, m, n,
                      steps_host, T, n_iter, interp_order, outval, state_dev, true, resume, out_dev);
}

static int members_convert(const void *src, void *dst, int n_members, int m, int n, bool to_disp) {
  PSH_REQUIRE_INIT();
  if (!src || !dst || n_members <= 0 || n_members > 65535 || m <= 0 || n <= 0)
    return psh::fail(PSH_EINVAL, "members state conversion: invalid argument");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * n;
  const dim3 grid(static_cast<unsigned>(std::min<size_t>((plane + 255) / 256, 4096)), n_members), block(256);
  if (to_disp) {
    hipLaunchKernelGGL(psh::members_state_to_disp, grid, block, 0, c.stream, static_cast<const uint4 *>(src), plane,
                       static_cast<double *>(dst));
  } else {
    hipLaunchKernelGGL(psh::members_disp_to_state, grid, block, 0, c.stream, static_cast<const double *>(src), plane,
                       static_cast<uint4 *>(dst));
  }
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_members_state_to_disp_dev(const void *state_dev, int n_members, int m, int n, double *disp_dev) {
  return members_convert(state_dev, disp_dev, n_members, m, n, true);
}

extern "C" int psh_members_disp_to_state_dev(const double *disp_dev, int n_members, int m, int n, void *state_dev) {
  return members_convert(disp_dev, state_dev, n_members, m, n, false);
}
