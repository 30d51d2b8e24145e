// k-nearest-neighbour inverse-distance-weighting of sparse motion vectors onto
// the full pixel grid, for gfx950.
//
// Replaces pysteps/utils/interpolate.py:26-114 (idwinterp2d) as it is called by
// dense_lucaskanade (pysteps/motion/lucaskanade.py:272-274):
//     for every pixel, the k nearest vectors (cKDTree.query, :80-86),
//     w = (d / res + dist_offset)^-power, normalised (:95-103), out = sum w*uv (:106-109).
// This is the dominant cost of the reference's dense LK (109 s at 4096^2).
//
// Design (docs/history.md 3.2 "idw_coarse + idw_fine3"): selection-bound FP32 VALU work, no MFMA, almost no
// HBM traffic (8 B written per pixel).
//  * one 256-thread workgroup per 16x16-pixel tile.  A tile-level pre-pass brackets the
//    k-th nearest distance R of the tile centre c from an LDS histogram of centre
//    distances (R_lo < R <= R_hi, one pass, LDS atomics).  With h the tile's half
//    diagonal every pixel's k-th neighbour distance lies in [R-h, R+h], hence vectors
//    beyond R_hi + 2h are pruned, vectors within R_lo - 2h are CERTAIN members of every
//    pixel's neighbourhood, and only the ring in between is undecided.  Candidates are
//    compacted in index order (deterministic summation order) into LDS as
//    [certain | undecided].
//  * each thread adds the certain vectors without any selection, then picks the
//    k - n_certain nearest of the ~10-15 undecided ones (LDS broadcast reads; unsorted
//    register set + running maximum, fully unrolled so it never leaves VGPRs; a
//    tile-uniform branch uses an 8-slot set when that suffices) and adds them.
//  * k >= L (use everything) and candidate overflow take a brute-force path.
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"

// The selection compares squared distances computed at several places of the kernel
// for EXACT equality (k-th smallest distance vs the candidates in the second sweep):
// every site must round identically, so implicit FMA contraction is off and the one
// fused form is written out in dist2().
#pragma clang fp contract(off)

namespace psh {
namespace {

__device__ __forceinline__ float dist2(float ax, float ay, float bx, float by) {
  const float dx = ax - bx, dy = ay - by;
  return fmaf(dx, dx, dy * dy);
}

constexpr int kTile = 16;      // 16x16 pixels per workgroup (32x32 measured slower: the wider halo
constexpr int kThreads = 256;  // adds ~35 % candidates per pixel, more than the pre-pass costs)
constexpr int kRowsPerPass = kThreads / kTile;
constexpr int kBins = 1024;  // centre-distance histogram: bin width = reach / 1024
constexpr int kCandCap = 512;  // candidates kept in LDS per tile (8 KiB, twice: raw + classified)

// (one copy of powf in the binary: inlined at every weight it made the fine pass 105 KB of code, most of it for a
// power nobody uses - the instruction cache is shared between the waves of two compute units)
__device__ __noinline__ float idw_pow_weight(float t, float power) { return powf(t, -power); }
__device__ __forceinline__ float idw_weight(float d, float power, float offset) {
  const float t = d + offset;
  // power 0.5 is the reference default: 1/sqrt(t) as one v_rsq_f32 (1 ulp)
  return power == 0.5f ? __builtin_amdgcn_rsqf(t) : idw_pow_weight(t, power);
}

// distance from its square: one v_sqrt_f32 (1 ulp) instead of the IEEE sequence
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// The k smallest squared distances seen so far, as an unsorted register set with a
// running maximum.  Only VALUES are kept: the maximum of the final set is the k-th
// smallest distance, and a second sweep over the (LDS-resident) candidates collects
// every candidate below it - half the registers and selects of an index-carrying set.
template <int KMAX>
struct KSmallest {
  float d2[KMAX];
  float worst;
  int worst_pos;

  __device__ __forceinline__ void refresh() {
    float w = -INFINITY;
    int wp = 0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (d2[j] > w) {
        w = d2[j];
        wp = j;
      }
    }
    worst = w;
    worst_pos = wp;
  }

  __device__ __forceinline__ void offer(float d) {
    if (d < worst) {
      float w = -INFINITY;
      int wp = 0;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        d2[j] = j == worst_pos ? d : d2[j];
        if (d2[j] > w) {
          w = d2[j];
          wp = j;
        }
      }
      worst = w;
      worst_pos = wp;
    }
  }
};

template <int KMAX>
struct TopK {
  float d2[KMAX];
  int idx[KMAX];
  float worst;
  int worst_pos;

  __device__ __forceinline__ void init(int k) {
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      // slots beyond k can never be the maximum, so they are never filled
      d2[j] = j < k ? INFINITY : -INFINITY;
      idx[j] = -1;
    }
    worst = INFINITY;
    worst_pos = 0;
  }

  // recompute the running maximum after slots were written directly
  __device__ __forceinline__ void refresh() {
    float w = -INFINITY;
    int wp = 0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (d2[j] > w) {
        w = d2[j];
        wp = j;
      }
    }
    worst = w;
    worst_pos = wp;
  }

  __device__ __forceinline__ void offer(float d, int i) {
    if (d < worst) {
      float w = -INFINITY;
      int wp = 0;
#pragma unroll
      for (int j = 0; j < KMAX; ++j) {
        const bool hit = j == worst_pos;
        d2[j] = hit ? d : d2[j];
        idx[j] = hit ? i : idx[j];
        if (d2[j] > w) {
          w = d2[j];
          wp = j;
        }
      }
      worst = w;
      worst_pos = wp;
    }
  }
};

// Pick the `need` nearest of the candidates cand[first .. first+count) for the pixel (px,py) and add
// their weighted values to the running sums.  Two sweeps over the (LDS-resident, broadcast-read)
// candidates: the first keeps the `need` smallest squared distances, the second collects every
// candidate strictly below the largest of them plus as many AT that distance (index order) as
// are needed.
template <int KS>
__device__ __forceinline__ void add_nearest(const float4 *cand, int first, int count, int need, float px,
                                            float py, float inv_res, float power, float offset,
                                            float &sw, float &su, float &sv) {
  KSmallest<KS> top;
  const int n_fill = min(need, count);
#pragma unroll
  for (int j = 0; j < KS; ++j) {
    top.d2[j] = -INFINITY;  // unused slots can never be the maximum
    if (j < n_fill) {
      const float4 c = cand[first + j];
      top.d2[j] = dist2(c.x, c.y, px, py);
    }
  }
  top.refresh();
  for (int i = n_fill; i < count; ++i) {
    const float4 c = cand[first + i];
    top.offer(dist2(c.x, c.y, px, py));
  }
  const float tau = top.worst;
  int below = 0;
#pragma unroll
  for (int j = 0; j < KS; ++j) below += (top.d2[j] > -INFINITY && top.d2[j] < tau) ? 1 : 0;
  int ties_wanted = n_fill - below;
  for (int i = 0; i < count; ++i) {
    const float4 c = cand[first + i];
    const float d2 = dist2(c.x, c.y, px, py);
    bool take = d2 < tau;
    if (d2 == tau && ties_wanted > 0) {
      take = true;
      --ties_wanted;
    }
    if (take) {
      const float w = idw_weight(fast_sqrt(d2) * inv_res, power, offset);
      sw += w;
      su += w * c.z;
      sv += w * c.w;
    }
  }
}

// Brute force over all L vectors, optional k selection (used for k >= L and overflow).
template <int KMAX>
__device__ __forceinline__ void idw_pixel_global(const float2 *__restrict__ xy,
                                                 const float2 *__restrict__ uv, int L, int k,
                                                 float px, float py, float inv_res, float power,
                                                 float offset, float &ou, float &ov) {
  float sw = 0.f, su = 0.f, sv = 0.f;
  if (k >= L) {
    for (int i = 0; i < L; ++i) {
      const float2 p = xy[i], val = uv[i];
      const float w = idw_weight(sqrtf(dist2(p.x, p.y, px, py)) * inv_res, power, offset);
      sw += w;
      su += w * val.x;
      sv += w * val.y;
    }
  } else {
    TopK<KMAX> top;
    top.init(k);
    for (int i = 0; i < L; ++i) {
      const float2 p = xy[i];
      top.offer(dist2(p.x, p.y, px, py), i);
    }
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (top.idx[j] >= 0) {
        const float2 val = uv[top.idx[j]];
        const float w = idw_weight(sqrtf(top.d2[j]) * inv_res, power, offset);
        sw += w;
        su += w * val.x;
        sv += w * val.y;
      }
    }
  }
  ou = su / sw;
  ov = sv / sw;
}

template <int KMAX>
__global__ __launch_bounds__(kThreads) void idw_knn(const float2 *__restrict__ xy,
                                                    const float2 *__restrict__ uv, int L, int k,
                                                    int m, int n, float x0, float dx_grid,
                                                    float y0, float dy_grid, float inv_res,
                                                    float power, float offset, float dmax,
                                                    float *__restrict__ out, int tiles_x,
                                                    int n_tiles, int tiles_per_xcd) {
  __shared__ int s_hist[kBins];
  __shared__ float4 s_raw[kCandCap];   // x, y, u, v of the pruned vectors, index order
  __shared__ float4 s_cand[kCandCap];  // the same, classified: [certain | undecided]
  __shared__ float s_radius, s_radius_lo;
  __shared__ int s_split[2];  // number of certain / undecided candidates

  const int b = blockIdx.x;
  const int tile = (b % kNumXcd) * tiles_per_xcd + b / kNumXcd;  // XCD-contiguous tiles
  if (tile >= n_tiles) return;
  const int tx = (tile % tiles_x) * kTile, ty = (tile / tiles_x) * kTile;
  const int tid = threadIdx.x;
  const size_t plane = static_cast<size_t>(m) * n;
  __shared__ int s_wave_count[4];

  int n_cand = 0;
  if (k < L) {
    // ---- tile pre-pass: radius holding >= k vectors around the tile centre --------
    const int wx = min(kTile, n - tx), wy = min(kTile, m - ty);
    const float cx = x0 + dx_grid * (static_cast<float>(tx) + 0.5f * static_cast<float>(wx - 1));
    const float cy = y0 + dy_grid * (static_cast<float>(ty) + 0.5f * static_cast<float>(wy - 1));
    const float hx = 0.5f * fabsf(dx_grid) * static_cast<float>(wx - 1);
    const float hy = 0.5f * fabsf(dy_grid) * static_cast<float>(wy - 1);
    const float half_diag = sqrtf(hx * hx + hy * hy);
    const float bin_w = dmax / static_cast<float>(kBins);

    for (int i = tid; i < kBins; i += kThreads) s_hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < L; i += kThreads) {
      const float2 p = xy[i];
      const float ddx = p.x - cx, ddy = p.y - cy;
      const int bin = min(static_cast<int>(sqrtf(ddx * ddx + ddy * ddy) / bin_w), kBins - 1);
      atomicAdd(&s_hist[bin], 1);
    }
    __syncthreads();
    if (tid < 64) {  // one wave: prefix over the bins (kBins/64 per lane), first bin reaching k
      constexpr int kPer = kBins / 64;
      int tot = 0;
      for (int q = 0; q < kPer; ++q) tot += s_hist[tid * kPer + q];
      int incl = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (tid >= d) incl += up;
      }
      int run = incl - tot;
      int first = kBins;
      for (int q = 0; q < kPer; ++q) {
        run += s_hist[tid * kPer + q];
        if (run >= k && first == kBins) first = tid * kPer + q;
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) first = min(first, __shfl_xor(first, d));
      // R_lo < (k-th smallest centre distance) <= R_hi; the last bin is open-ended
      if (tid == 0) {
        s_radius = first >= kBins - 1 ? INFINITY : static_cast<float>(first + 1) * bin_w;
        s_radius_lo = static_cast<float>(min(first, kBins - 1)) * bin_w;
      }
    }
    __syncthreads();
    const float reach = s_radius + 2.f * half_diag + 1e-3f * (s_radius + half_diag);
    // ---- ordered compaction of the candidates: each wave owns a contiguous quarter of
    // the vectors, counts first, then writes behind the waves before it (index order is
    // kept, so the summation order is deterministic)
    const int wave = tid >> 6, lane = tid & 63;
    const int per_wave = (L + 3) / 4;
    const int i_begin = wave * per_wave, i_end = min(L, i_begin + per_wave);
    auto wanted = [&](int i, float2 &p) {
      if (i >= i_end) return false;
      p = xy[i];
      const float ddx = p.x - cx, ddy = p.y - cy;
      return sqrtf(ddx * ddx + ddy * ddy) <= reach;
    };
    int mine = 0;
    for (int i0 = i_begin; i0 < i_end; i0 += 64) {
      float2 p;
      mine += __popcll(__ballot(wanted(i0 + lane, p)));
    }
    if (lane == 0) s_wave_count[wave] = mine;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_wave_count[w];
    n_cand = s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
    if (n_cand <= kCandCap) {
      for (int i0 = i_begin; i0 < i_end; i0 += 64) {
        float2 p = make_float2(0.f, 0.f);
        const bool keep = wanted(i0 + lane, p);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
          const float2 val = uv[i0 + lane];
          s_raw[base + __popcll(mask & ((1ull << lane) - 1ull))] = make_float4(p.x, p.y, val.x, val.y);
        }
        base += __popcll(mask);
      }
    }
    __syncthreads();
    // ---- classification against the bracket R_lo < R <= R_hi of the k-th centre distance ------
    // For a pixel p of the tile (|p - c| <= h) its k-th neighbour distance lies in [R-h, R+h], so a
    // candidate with centre distance <= R_lo - 2h belongs to every pixel's k nearest ("certain")
    // and one beyond R_hi + 2h (already pruned) to none; only the ring in between needs a
    // per-pixel selection, of k - n_certain vectors.  Wave 0 orders the candidates
    // [certain | undecided], each group in index order.
    if (n_cand <= kCandCap) {
      if (tid < 64) {
        const float sure_below = s_radius_lo - 2.f * half_diag - 1e-3f * (s_radius_lo + half_diag);
        int n_sure = 0, n_ring = 0;
        for (int pass = 0; pass < 2; ++pass) {
          int at = pass == 0 ? 0 : n_sure;
          for (int i0 = 0; i0 < n_cand; i0 += 64) {
            const int i = i0 + tid;
            bool keep = false;
            float4 cnd = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n_cand) {
              cnd = s_raw[i];
              const float ddx = cnd.x - cx, ddy = cnd.y - cy;
              const bool sure = sqrtf(ddx * ddx + ddy * ddy) <= sure_below;
              keep = pass == 0 ? sure : !sure;
            }
            const unsigned long long mask = __ballot(keep);
            PSH_DASSERT(!keep || at + __popcll(mask & ((1ull << tid) - 1ull)) < kCandCap);
            if (keep) s_cand[at + __popcll(mask & ((1ull << tid) - 1ull))] = cnd;
            at += __popcll(mask);
          }
          if (pass == 0) n_sure = at; else n_ring = at - n_sure;
        }
        if (tid == 0) {
          s_split[0] = n_sure;
          s_split[1] = n_ring;
        }
      }
      __syncthreads();
    }
  }
  const int n_sure = s_split[0], n_ring = s_split[1];

  for (int q = 0; q < kTile / kRowsPerPass; ++q) {
    const int ix = tx + (tid % kTile), iy = ty + (tid / kTile) + q * kRowsPerPass;
    if (ix >= n || iy >= m) continue;
    const float px = x0 + dx_grid * static_cast<float>(ix);
    const float py = y0 + dy_grid * static_cast<float>(iy);
    float ou, ov;
    if (k >= L || n_cand > kCandCap) {
      // no selection at all, or pathological clustering: exact brute force
      idw_pixel_global<KMAX>(xy, uv, L, k, px, py, inv_res, power, offset, ou, ov);
    } else {
      float sw = 0.f, su = 0.f, sv = 0.f;
      for (int i = 0; i < n_sure; ++i) {  // in every pixel's neighbourhood: no selection
        const float4 c = s_cand[i];    // same address in every lane: LDS broadcast
        const float w = idw_weight(fast_sqrt(dist2(c.x, c.y, px, py)) * inv_res, power, offset);
        sw += w;
        su += w * c.z;
        sv += w * c.w;
      }
      const int need = k - n_sure;  // >= 1: fewer than k vectors lie strictly inside R
      if (need <= 8) {              // tile-uniform branch: small selection sets are much cheaper
        add_nearest<8>(s_cand, n_sure, n_ring, need, px, py, inv_res, power, offset, sw, su, sv);
      } else {
        add_nearest<KMAX>(s_cand, n_sure, n_ring, need, px, py, inv_res, power, offset, sw, su, sv);
      }
      ou = su / sw;
      ov = sv / sw;
    }
    out[static_cast<size_t>(iy) * n + ix] = ou;
    out[plane + static_cast<size_t>(iy) * n + ix] = ov;
  }
}

// ---- two-level variant (default) ----------------------------------------------------------
// The per-tile pre-pass above scans all L vectors for every 16x16 tile: at L ~ 1000 that is a
// third of the kernel, and the ring of undecided vectors is as thick as four half-diagonals of
// the tile.  Here the scan over all L is done once per 64x64 SUPERTILE (idw_coarse: the same
// bracket and ordered compaction, into a global list of typically 30-60 vectors), and the
// fine pass runs one WAVE per small tile (no workgroup barriers that span waves): it brackets the k-th centre distance of its own tile from the supertile's list, so
// the ring is thinner and fewer vectors need the per-pixel selection.
// Exactness: a pixel q of a supertile with centre c and half-diagonal H has its k-th neighbour
// within R_hi + H of q, hence within R_hi + 2H of c - the supertile list holds every vector any
// of its pixels can need, and a fine tile's k-th centre distance computed from the list is the
// true one (vectors outside the list are farther than R_hi + H from any fine centre).
constexpr int kSuper = 64;      // supertile edge in pixels
constexpr int kSuperCap = 256;  // vectors kept per supertile (4 KiB of float4)
constexpr int kFineCap = 96;    // vectors per fine tile in LDS

struct SuperHeader {
  int count;        // vectors in the list, or > kSuperCap: overflow (brute force in the fine pass)
  float reach;      // the list holds every vector within `reach` of the supertile centre
  float half_diag;  // of the supertile
  float pad;
};

__global__ __launch_bounds__(kThreads) void idw_coarse(const float2 *__restrict__ xy,
                                                       const float2 *__restrict__ uv, int L, int k, int m,
                                                       int n, float x0, float dx_grid, float y0,
                                                       float dy_grid, float dmax, int supers_x,
                                                       SuperHeader *__restrict__ headers,
                                                       float4 *__restrict__ lists,
                                                       const IdwDyn *__restrict__ dyn) {
  __shared__ int s_hist[kBins];
  __shared__ float s_radius;
  __shared__ int s_wave_count[4];
  const int sup = blockIdx.x;
  if (dyn) {  // sample count, neighbour count and reach live in device memory (common.h IdwDyn)
    L = dyn->L;
    dmax = dyn->reach;
    if (dyn->mode != 0 || k >= L) {  // constant field / every sample is a neighbour: no lists
      if (threadIdx.x == 0) headers[sup] = SuperHeader{0, 0.f, 0.f, 0.f};
      return;
    }
  }
  const int tx = (sup % supers_x) * kSuper, ty = (sup / supers_x) * kSuper;
  const int tid = threadIdx.x;
  const int wx = min(kSuper, n - tx), wy = min(kSuper, m - ty);
  const float cx = x0 + dx_grid * (static_cast<float>(tx) + 0.5f * static_cast<float>(wx - 1));
  const float cy = y0 + dy_grid * (static_cast<float>(ty) + 0.5f * static_cast<float>(wy - 1));
  const float hx = 0.5f * fabsf(dx_grid) * static_cast<float>(wx - 1);
  const float hy = 0.5f * fabsf(dy_grid) * static_cast<float>(wy - 1);
  const float half_diag = sqrtf(hx * hx + hy * hy);
  // (a multiplication instead of the division by the bin width: a vector that lands in the neighbouring bin
  // moves the bracket by one bin at most, and `reach` carries a 1e-3 margin; the lists stay supersets of what
  // the fine pass needs, in index order - the field does not change)
  const float bin_w = dmax / static_cast<float>(kBins), inv_bin_w = static_cast<float>(kBins) / dmax;
  for (int i = tid; i < kBins; i += kThreads) s_hist[i] = 0;
  __syncthreads();
  // each wave owns a contiguous quarter of the vectors (ordered compaction below) and keeps its vectors'
  // positions and squared centre distances in registers: the list is read once
  const int wave = tid >> 6, lane = tid & 63;
  const int per_wave = (L + 3) / 4;
  const int i_begin = wave * per_wave, i_end = min(L, i_begin + per_wave);
  constexpr int kKeep = 8;  // 64 x 8 vectors per wave in registers (L <= 2048); longer lists re-read the rest
  float2 pk[kKeep];
  float d2k[kKeep];
#pragma unroll
  for (int q = 0; q < kKeep; ++q) {
    const int i = i_begin + q * 64 + lane;
    d2k[q] = INFINITY;
    pk[q] = make_float2(0.f, 0.f);
    if (i < i_end) {
      pk[q] = xy[i];
      const float ddx = pk[q].x - cx, ddy = pk[q].y - cy;
      d2k[q] = ddx * ddx + ddy * ddy;
      atomicAdd(&s_hist[min(static_cast<int>(sqrtf(d2k[q]) * inv_bin_w), kBins - 1)], 1);
    }
  }
  for (int i = i_begin + kKeep * 64 + lane; i < i_end; i += 64) {
    const float2 p = xy[i];
    const float ddx = p.x - cx, ddy = p.y - cy;
    atomicAdd(&s_hist[min(static_cast<int>(sqrtf(ddx * ddx + ddy * ddy) * inv_bin_w), kBins - 1)], 1);
  }
  __syncthreads();
  if (tid < 64) {
    constexpr int kPer = kBins / 64;
    int tot = 0;
    for (int q = 0; q < kPer; ++q) tot += s_hist[tid * kPer + q];
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (tid >= d) incl += up;
    }
    int run = incl - tot;
    int first = kBins;
    for (int q = 0; q < kPer; ++q) {
      run += s_hist[tid * kPer + q];
      if (run >= k && first == kBins) first = tid * kPer + q;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) first = min(first, __shfl_xor(first, d));
    if (tid == 0) s_radius = first >= kBins - 1 ? INFINITY : static_cast<float>(first + 1) * bin_w;
  }
  __syncthreads();
  const float reach = s_radius + 2.f * half_diag + 1e-3f * (s_radius + half_diag);
  const float reach2 = reach * reach;
  auto wanted_tail = [&](int i, float2 &p) {
    if (i >= i_end) return false;
    p = xy[i];
    const float ddx = p.x - cx, ddy = p.y - cy;
    return ddx * ddx + ddy * ddy <= reach2;
  };
  unsigned long long maskk[kKeep];
  int mine = 0;
#pragma unroll
  for (int q = 0; q < kKeep; ++q) {
    maskk[q] = __ballot(d2k[q] <= reach2);
    mine += __popcll(maskk[q]);
  }
  for (int i0 = i_begin + kKeep * 64; i0 < i_end; i0 += 64) {
    float2 p;
    mine += __popcll(__ballot(wanted_tail(i0 + lane, p)));
  }
  if (lane == 0) s_wave_count[wave] = mine;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += s_wave_count[w];
  const int n_cand = s_wave_count[0] + s_wave_count[1] + s_wave_count[2] + s_wave_count[3];
  if (tid == 0) {
    SuperHeader h;
    h.count = n_cand;
    h.reach = reach;
    h.half_diag = half_diag;
    h.pad = 0.f;
    headers[sup] = h;
  }
  if (n_cand > kSuperCap) return;
  float4 *dst = lists + static_cast<size_t>(sup) * kSuperCap;
#pragma unroll
  for (int q = 0; q < kKeep; ++q) {
    if (maskk[q] == 0ull) continue;  // (uniform)
    if ((maskk[q] >> lane) & 1ull) {
      const float2 val = uv[i_begin + q * 64 + lane];
      dst[base + __popcll(maskk[q] & ((1ull << lane) - 1ull))] = make_float4(pk[q].x, pk[q].y, val.x, val.y);
    }
    base += __popcll(maskk[q]);
  }
  for (int i0 = i_begin + kKeep * 64; i0 < i_end; i0 += 64) {
    float2 p = make_float2(0.f, 0.f);
    const bool keep = wanted_tail(i0 + lane, p);
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const float2 val = uv[i0 + lane];
      dst[base + __popcll(mask & ((1ull << lane) - 1ull))] = make_float4(p.x, p.y, val.x, val.y);
    }
    base += __popcll(mask);
  }
}

// ---- fine pass: one wave per 16 x 8 tile, two pixels per lane ----------------------------------
// The supertile lists are short (30-60 vectors), so the k-th smallest centre distance of the tile is
// found EXACTLY - and without LDS - by bisection on the bit patterns of the squared distances (a
// non-negative float orders like its bits): 25 rounds of one compare + ballot + scalar popcount fix
// the value up to its 6 lowest mantissa bits.  Squared distances are classified against squared
// radii (no square root per vector), the certain and the undecided vectors are compacted into LDS with
// one ballot each, and a small ring (<= 8 vectors, the rule at the reference's density) is ranked
// pairwise in registers instead of going through the replace-the-maximum set.  Per pixel: dist2, the
// weight, the sums in the order "certain vectors in index order, then the chosen ring vectors in index
// order".
// The pass is bound by instruction issue (counters of the one-pixel-per-lane forms of rounds 2-4,
// profiles/r04/d_idw_fine_pmc.csv, deleted after the comparison in e_idw_ab.txt): one wave-wide VALU
// instruction occupies a SIMD for four cycles on this chip whatever its type, the sum loop is a chain
// of dependent operations behind an LDS read, and every wave repeats the tile prologue.  So a wave owns
// a 16 x 8 tile and a lane the two pixels (x, y), (x + 8, y): the per-vector arithmetic of the pair
// runs in packed FP32 instructions (v_pk_add / v_pk_mul / v_pk_fma_f32: IEEE per component, each
// pixel's operations and their order unchanged), only the two transcendentals stay scalar; two vectors
// are in flight per loop iteration (four independent chains), and the prologue is paid once per 128
// pixels.  Against an 8 x 8 tile the half diagonal grows from 4.9 to 8.3 pixels, so the undecided ring
// holds ~1.7 times as many vectors - small against what the packing saves.
typedef float float2v __attribute__((ext_vector_type(2)));
constexpr int kFineW = 16, kFineH = 8;

template <bool HALF>  // HALF: power == 0.5, the reference default (one v_rsq_f32 per weight)
__device__ __forceinline__ void idw_accumulate2(const float4 c, const float2v px, float py, float inv_res, float power,
                                                float offset, float2v &sw, float2v &su, float2v &sv) {
  const float2v dx = float2v{c.x, c.x} - px;
  const float dy = c.y - py;
  const float dy2 = dy * dy;
  const float2v d2 = __builtin_elementwise_fma(dx, dx, float2v{dy2, dy2});  // = dist2() per component
  const float2v root = float2v{fast_sqrt(d2.x), fast_sqrt(d2.y)};
  float2v w;
  if constexpr (HALF) {
    // (d / res + offset and the two weighted sums as fused multiply-adds: three packed instructions less per
    // vector than the multiply-then-add forms - every wave64 instruction costs the SIMD four cycles here - and
    // one rounding less each; nothing in the kernel compares these values for equality)
    const float2v t = __builtin_elementwise_fma(root, float2v{inv_res, inv_res}, float2v{offset, offset});
    w = float2v{__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y)};
  } else {
    const float2v d = root * inv_res;
    w = float2v{idw_weight(d.x, power, offset), idw_weight(d.y, power, offset)};
  }
  sw += w;
  su = __builtin_elementwise_fma(w, float2v{c.z, c.z}, su);
  sv = __builtin_elementwise_fma(w, float2v{c.w, c.w}, sv);
}

// the `need` nearest of a ring of NR <= 8 vectors: every member ranks itself among the others (ties: lower index
// first, as the second sweep of add_nearest takes them) - for the lane's TWO pixels: distances and weights in packed instructions, the ranks per component (a
// compare has no packed form); a vector that is not among a pixel's `need` nearest enters its sums with weight zero.
// One instantiation per ring size (the ring is tile-uniform): NR (NR - 1) comparisons per pixel instead of the 56 of
// an eight-slot ring padded with +inf - a ring holds four or five vectors as a rule.
template <bool HALF, int NR>
__device__ __forceinline__ void idw_small_ring2(const float4 *ring, int need, const float2v px, float py,
                                                float inv_res, float power, float offset, float2v &sw, float2v &su,
                                                float2v &sv) {
  float2v d2[NR];
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const float4 c = ring[j];
    const float2v dx = float2v{c.x, c.x} - px;
    const float dy = c.y - py;
    const float dy2 = dy * dy;
    d2[j] = __builtin_elementwise_fma(dx, dx, float2v{dy2, dy2});  // = dist2() per component
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    int rank_a = 0, rank_b = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i == j) continue;
      // (ties go to the lower index)
      rank_a += (i < j ? d2[i].x <= d2[j].x : d2[i].x < d2[j].x) ? 1 : 0;
      rank_b += (i < j ? d2[i].y <= d2[j].y : d2[i].y < d2[j].y) ? 1 : 0;
    }
    const float4 c = ring[j];
    const float2v root = float2v{fast_sqrt(d2[j].x), fast_sqrt(d2[j].y)};
    float2v w;
    if constexpr (HALF) {
      const float2v t = __builtin_elementwise_fma(root, float2v{inv_res, inv_res}, float2v{offset, offset});
      w = float2v{__builtin_amdgcn_rsqf(t.x), __builtin_amdgcn_rsqf(t.y)};
    } else {
      const float2v d = root * inv_res;
      w = float2v{idw_weight(d.x, power, offset), idw_weight(d.y, power, offset)};
    }
    w = float2v{rank_a < need ? w.x : 0.f, rank_b < need ? w.y : 0.f};
    sw += w;
    su = __builtin_elementwise_fma(w, float2v{c.z, c.z}, su);
    sv = __builtin_elementwise_fma(w, float2v{c.w, c.w}, sv);
  }
}

template <bool HALF>
__device__ __forceinline__ void idw_small_ring2(const float4 *ring, int n_ring, int need, const float2v px, float py,
                                                float inv_res, float power, float offset, float2v &sw, float2v &su,
                                                float2v &sv) {
  switch (n_ring) {  // (uniform)
#define PSH_RING_CASE(NR)                                                                      \
  case NR:                                                                                     \
    idw_small_ring2<HALF, NR>(ring, need, px, py, inv_res, power, offset, sw, su, sv);         \
    break;
    PSH_RING_CASE(1)
    PSH_RING_CASE(2)
    PSH_RING_CASE(3)
    PSH_RING_CASE(4)
    PSH_RING_CASE(5)
    PSH_RING_CASE(6)
    PSH_RING_CASE(7)
    PSH_RING_CASE(8)
#undef PSH_RING_CASE
    default:
      break;
  }
}

template <int KMAX>
__global__ __launch_bounds__(64) void idw_fine3(const float2 *__restrict__ xy, const float2 *__restrict__ uv,
                                                int L, int k, int m, int n, float x0, float dx_grid,
                                                float y0, float dy_grid, float inv_res, float power,
                                                float offset, float *__restrict__ out, int supers_x,
                                                const SuperHeader *__restrict__ headers,
                                                const float4 *__restrict__ lists, int tiles_x, int n_tiles,
                                                int tiles_per_xcd, const IdwDyn *__restrict__ dyn,
                                                float2 *__restrict__ out_uv) {
  // out_uv (may be nullptr): the same field once more as {u, v} pairs per pixel - the layout the extrapolator
  // gathers the motion field from (semilag.hip pack_velocity), written here instead of by a pass of its own
  __shared__ float4 s_cand[kFineCap];  // [certain | undecided], each group in index order
  const int b = blockIdx.x;
  const int tile = (b % kNumXcd) * tiles_per_xcd + b / kNumXcd;  // XCD-contiguous tiles
  if (tile >= n_tiles) return;
  const int tx = (tile % tiles_x) * kFineW, ty = (tile / tiles_x) * kFineH;
  const int lane = threadIdx.x;
  const int ix_a = tx + (lane & 7), ix_b = ix_a + 8, iy = ty + (lane >> 3);
  const bool live_a = ix_a < n && iy < m, live_b = ix_b < n && iy < m;
  const size_t plane = static_cast<size_t>(m) * n;
  const size_t at_a = static_cast<size_t>(iy) * n + ix_a;
  if (dyn) {
    L = dyn->L;
    k = min(k, L);
    if (dyn->mode != 0) {  // the interpolator's trivial cases (decorators.py:199-208): constant field
      const float cu = dyn->cu, cv = dyn->cv;
      if (live_a) {
        out[at_a] = cu;
        out[plane + at_a] = cv;
        if (out_uv) out_uv[at_a] = make_float2(cu, cv);
      }
      if (live_b) {
        out[at_a + 8] = cu;
        out[plane + at_a + 8] = cv;
        if (out_uv) out_uv[at_a + 8] = make_float2(cu, cv);
      }
      return;
    }
  }
  const float2v px = {x0 + dx_grid * static_cast<float>(ix_a), x0 + dx_grid * static_cast<float>(ix_b)};
  const float py = y0 + dy_grid * static_cast<float>(iy);
  const int sup = (ty / kSuper) * supers_x + tx / kSuper;
  const SuperHeader hdr = headers[sup];
  const float4 *list = lists + static_cast<size_t>(sup) * kSuperCap;
  const int n_s = __builtin_amdgcn_readfirstlane(hdr.count);

  bool brute = k >= L || n_s > kSuperCap || n_s < k;
  int n_sure = 0, n_ring = 0;
  if (!brute) {
    const int wx = min(kFineW, n - tx), wy = min(kFineH, m - ty);
    const float cx = x0 + dx_grid * (static_cast<float>(tx) + 0.5f * static_cast<float>(wx - 1));
    const float cy = y0 + dy_grid * (static_cast<float>(ty) + 0.5f * static_cast<float>(wy - 1));
    const float hx = 0.5f * fabsf(dx_grid) * static_cast<float>(wx - 1);
    const float hy = 0.5f * fabsf(dy_grid) * static_cast<float>(wy - 1);
    const float half_diag = sqrtf(hx * hx + hy * hy);
    constexpr int kPerLane = kSuperCap / 64;
    const int chunks = (n_s + 63) >> 6;  // (uniform)
    float4 c[kPerLane];
    unsigned key[kPerLane];  // bits of the squared centre distance; missing entries: +inf
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) {
      key[j] = 0x7f800000u;
      c[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < chunks) {
        const int i = j * 64 + lane;
        if (i < n_s) {
          c[j] = list[i];
          key[j] = __float_as_uint(dist2(c[j].x, c[j].y, cx, cy));
        }
      }
    }
    // k-th smallest key: the largest T with fewer than k keys below it, built from the top bit
    // (the 6 lowest mantissa bits stay open: T <= k-th smallest squared distance < T + 64 ulp)
    unsigned T = 0u;
    if (chunks == 1) {
#pragma unroll
      for (int bit = 30; bit >= 6; --bit) {
        const unsigned t = T | (1u << bit);
        if (__popcll(__ballot(key[0] < t)) < k) T = t;
      }
    } else {
#pragma unroll 1
      for (int bit = 30; bit >= 6; --bit) {
        const unsigned t = T | (1u << bit);
        int below = 0;
#pragma unroll
        for (int j = 0; j < kPerLane; ++j)
          if (j < chunks) below += __popcll(__ballot(key[j] < t));
        if (below < k) T = t;
      }
    }
    const float r_lo = fast_sqrt(__uint_as_float(T)) * (1.f - 1e-6f);
    const float r_hi = fast_sqrt(__uint_as_float(T + 64u)) * (1.f + 1e-6f);
    const float reach = r_hi + 2.f * half_diag + 1e-3f * (r_hi + half_diag);
    const float sure_below = r_lo - 2.f * half_diag - 1e-3f * (r_lo + half_diag);
    const float reach2 = reach * reach;
    const float sure2 = sure_below > 0.f ? sure_below * sure_below : -1.f;
    // certain vectors first, then the undecided ring, each in list (= index) order
    unsigned long long sure_mask[kPerLane], ring_mask[kPerLane];
    int tot_sure = 0, tot_ring = 0;
#pragma unroll
    for (int j = 0; j < kPerLane; ++j) {
      sure_mask[j] = ring_mask[j] = 0ull;
      if (j < chunks) {
        const float d2 = __uint_as_float(key[j]);
        const bool sure = d2 <= sure2;
        sure_mask[j] = __ballot(sure);
        ring_mask[j] = __ballot(!sure && d2 <= reach2);
        tot_sure += __popcll(sure_mask[j]);
        tot_ring += __popcll(ring_mask[j]);
      }
    }
    n_sure = tot_sure;
    n_ring = tot_ring;
    brute = n_sure + n_ring > kFineCap;  // pathological clustering: exact brute force
    if (!brute) {
      const unsigned long long lt = (1ull << lane) - 1ull;
      int at_sure = 0, at_ring = n_sure;
#pragma unroll
      for (int j = 0; j < kPerLane; ++j) {
        if (j < chunks) {
          const bool sure = (sure_mask[j] >> lane) & 1ull, ring = (ring_mask[j] >> lane) & 1ull;
          const int slot = sure ? at_sure + __popcll(sure_mask[j] & lt) : at_ring + __popcll(ring_mask[j] & lt);
          PSH_DASSERT(!(sure || ring) || (slot >= 0 && slot < kFineCap));  // the classified list fits its LDS block
          if (sure || ring) s_cand[slot] = c[j];
          at_sure += __popcll(sure_mask[j]);
          at_ring += __popcll(ring_mask[j]);
        }
      }
    }
    __syncthreads();  // (one wave: orders the LDS writes before the broadcast reads)
  }
  if (!live_a) return;  // (live_b implies live_a)
  float2v o_u, o_v;
  if (brute) {
    float ua, va, ub = 0.f, vb = 0.f;
    idw_pixel_global<KMAX>(xy, uv, L, k, px.x, py, inv_res, power, offset, ua, va);
    if (live_b) idw_pixel_global<KMAX>(xy, uv, L, k, px.y, py, inv_res, power, offset, ub, vb);
    o_u = float2v{ua, ub};
    o_v = float2v{va, vb};
  } else {
    float2v sw = {0.f, 0.f}, su = {0.f, 0.f}, sv = {0.f, 0.f};
    // the certain vectors - in every pixel's neighbourhood, no selection - two per iteration
    auto sum_certain = [&](auto half) {
      constexpr bool kHalf = decltype(half)::value;
      int i = 0;
      for (; i + 1 < n_sure; i += 2) {
        const float4 c0 = s_cand[i], c1 = s_cand[i + 1];  // same address in every lane: LDS broadcast
        idw_accumulate2<kHalf>(c0, px, py, inv_res, power, offset, sw, su, sv);
        idw_accumulate2<kHalf>(c1, px, py, inv_res, power, offset, sw, su, sv);
      }
      if (i < n_sure) idw_accumulate2<kHalf>(s_cand[i], px, py, inv_res, power, offset, sw, su, sv);
    };
    if (power == 0.5f) {
      sum_certain(std::true_type{});
    } else {
      sum_certain(std::false_type{});
    }
    const int need = k - n_sure;  // >= 1: fewer than k vectors lie strictly inside R_lo
    // the ring: a small one (the rule) for both pixels at once, else pixel by pixel (scalar copies of the sums:
    // vector elements cannot be passed by reference)
    if (n_ring <= 8) {
      if (power == 0.5f) {
        idw_small_ring2<true>(s_cand + n_sure, n_ring, need, px, py, inv_res, power, offset, sw, su, sv);
      } else {
        idw_small_ring2<false>(s_cand + n_sure, n_ring, need, px, py, inv_res, power, offset, sw, su, sv);
      }
    }
    float wa = sw.x, ua = su.x, va = sv.x, wb = sw.y, ub = su.y, vb = sv.y;
    if (n_ring <= 8) {
    } else if (need <= 8) {
      add_nearest<8>(s_cand, n_sure, n_ring, need, px.x, py, inv_res, power, offset, wa, ua, va);
      add_nearest<8>(s_cand, n_sure, n_ring, need, px.y, py, inv_res, power, offset, wb, ub, vb);
    } else {
      add_nearest<KMAX>(s_cand, n_sure, n_ring, need, px.x, py, inv_res, power, offset, wa, ua, va);
      add_nearest<KMAX>(s_cand, n_sure, n_ring, need, px.y, py, inv_res, power, offset, wb, ub, vb);
    }
    sw = float2v{wa, wb};
    su = float2v{ua, ub};
    sv = float2v{va, vb};
    // (one reciprocal per pixel - v_rcp_f32, 1 ulp - and two products instead of two IEEE divisions: ~40 of the
    // ~700 instructions a tile costs; the sums themselves carry more rounding than that)
    const float2v inv = float2v{__builtin_amdgcn_rcpf(sw.x), __builtin_amdgcn_rcpf(sw.y)};
    o_u = su * inv;
    o_v = sv * inv;
  }
  out[at_a] = o_u.x;
  out[plane + at_a] = o_v.x;
  if (out_uv) out_uv[at_a] = make_float2(o_u.x, o_v.x);
  if (live_b) {
    out[at_a + 8] = o_u.y;
    out[plane + at_a + 8] = o_v.y;
    if (out_uv) out_uv[at_a + 8] = make_float2(o_u.y, o_v.y);
  }
}

}  // namespace

// 0 = two-level (default), 1 = one pre-pass per 16x16 tile (kept as the independent second
// implementation tests/test_idw_gpu.py compares with)
static int g_idw_variant = [] {
  const char *e = std::getenv("PYSTEPS_HIP_IDW_VARIANT");
  return e ? std::atoi(e) : 0;
}();
void set_idw_variant(int v) { g_idw_variant = v; }

size_t idw_scratch_bytes(int m, int n) {
  const size_t supers = static_cast<size_t>((n + kSuper - 1) / kSuper) * ((m + kSuper - 1) / kSuper);
  return supers * (sizeof(SuperHeader) + kSuperCap * sizeof(float4));
}

hipError_t launch_idw(const IdwArgs &a, hipStream_t stream) {
  const float2 *xy = reinterpret_cast<const float2 *>(a.xy);
  const float2 *uv = reinterpret_cast<const float2 *>(a.uv);
  // (with a device-resident sample count the instantiation follows the requested k: should the
  // samples turn out to be fewer, every instantiation takes the same all-samples path)
  const int k_eff = a.dyn ? (a.k > 32 ? 1 : a.k) : a.k >= a.L ? 1 : a.k;
  if ((g_idw_variant != 1 || a.dyn != nullptr) && a.scratch != nullptr) {
    const int supers_x = (a.n + kSuper - 1) / kSuper, supers_y = (a.m + kSuper - 1) / kSuper;
    const int n_super = supers_x * supers_y;
    SuperHeader *headers = static_cast<SuperHeader *>(a.scratch);
    float4 *lists = reinterpret_cast<float4 *>(headers + n_super);
    if (a.dyn != nullptr || a.k < a.L) {
      hipLaunchKernelGGL(idw_coarse, dim3(n_super), dim3(kThreads), 0, stream, xy, uv, a.L, a.k, a.m, a.n, a.x0,
                         a.dx, a.y0, a.dy, a.dmax, supers_x, headers, lists, a.dyn);
    } else {
      hipError_t e = hipMemsetAsync(headers, 0, n_super * sizeof(SuperHeader), stream);
      if (e != hipSuccess) return e;
    }
    const int tiles_x = (a.n + kFineW - 1) / kFineW, tiles_y = (a.m + kFineH - 1) / kFineH;
    const int n_tiles = tiles_x * tiles_y;
    const int tiles_per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
    const dim3 grid(tiles_per_xcd * kNumXcd), block(64);
#define PSH_IDW_FINE(KMAX)                                                                                           \
  hipLaunchKernelGGL((idw_fine3<KMAX>), grid, block, 0, stream, xy, uv, a.L, a.k, a.m, a.n, a.x0, a.dx, a.y0, a.dy, \
                     a.inv_res, a.power, a.offset, a.out, supers_x, headers, lists, tiles_x, n_tiles, tiles_per_xcd, a.dyn, \
                     reinterpret_cast<float2 *>(a.out_uv))
    if (k_eff <= 8) {
      PSH_IDW_FINE(8);
    } else if (k_eff <= 20) {
      PSH_IDW_FINE(20);
    } else {
      PSH_IDW_FINE(32);
    }
#undef PSH_IDW_FINE
    return hipGetLastError();
  }
  if (a.out_uv) return hipErrorInvalidValue;  // (only the two-level kernels write the interleaved copy)
  const int tiles_x = (a.n + kTile - 1) / kTile;
  const int tiles_y = (a.m + kTile - 1) / kTile;
  const int n_tiles = tiles_x * tiles_y;
  const int tiles_per_xcd = (n_tiles + kNumXcd - 1) / kNumXcd;
  const dim3 grid(tiles_per_xcd * kNumXcd), block(kThreads);
#define PSH_IDW_LAUNCH(KMAX)                                                                    \
  hipLaunchKernelGGL((idw_knn<KMAX>), grid, block, 0, stream, xy, uv, a.L, a.k, a.m, a.n, a.x0, \
                     a.dx, a.y0, a.dy, a.inv_res, a.power, a.offset, a.dmax, a.out, tiles_x,    \
                     n_tiles, tiles_per_xcd)
  if (k_eff <= 8) {
    PSH_IDW_LAUNCH(8);
  } else if (k_eff <= 20) {
    PSH_IDW_LAUNCH(20);
  } else {
    PSH_IDW_LAUNCH(32);
  }
#undef PSH_IDW_LAUNCH
  return hipGetLastError();
}

}  // namespace psh

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
static int idw_check(int L, int k, int m, int n, double dx, double dy, double power) {
  if (L <= 0) return psh::fail(PSH_EINVAL, "idw: need at least one sample (L=%d)", L);
  if (m <= 0 || n <= 0) return psh::fail(PSH_EINVAL, "idw: invalid grid (%d,%d)", m, n);
  if (k <= 0) return psh::fail(PSH_EINVAL, "idw: k must be positive (got %d)", k);
  if (k < L && k > 32)
    return psh::fail(PSH_EUNSUPPORTED, "idw: k=%d > 32 with k < L is not implemented", k);
  if (dx == 0.0 || dy == 0.0) return psh::fail(PSH_EINVAL, "idw: zero grid spacing");
  if (!(power > 0.0)) return psh::fail(PSH_EINVAL, "idw: power must be positive");
  return PSH_OK;
}

extern "C" int psh_idw_dev(const float *xy_dev, const float *values_dev, int L, int m, int n,
                           double x0, double dx, double y0, double dy, int k, double power,
                           double dist_offset, double reach_hint, float *out_dev) {
  PSH_REQUIRE_INIT();
  if (int rc = idw_check(L, k, m, n, dx, dy, power)) return rc;
  if (!xy_dev || !values_dev || !out_dev) return psh::fail(PSH_EINVAL, "idw: NULL pointer");
  if (!(reach_hint > 0.0)) return psh::fail(PSH_EINVAL, "idw: reach_hint must be positive");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::IdwArgs a;
  a.xy = xy_dev;
  a.uv = values_dev;
  a.out = out_dev;
  a.L = L;
  a.k = k < L ? k : L;
  a.m = m;
  a.n = n;
  a.x0 = static_cast<float>(x0);
  a.dx = static_cast<float>(dx);
  a.y0 = static_cast<float>(y0);
  a.dy = static_cast<float>(dy);
  // distances are expressed in pixels: mean grid resolution (interpolate.py:89-93)
  const double res = 0.5 * (fabs(dx) + fabs(dy));
  a.inv_res = static_cast<float>(1.0 / res);
  a.power = static_cast<float>(power);
  a.offset = static_cast<float>(dist_offset);
  a.dmax = static_cast<float>(reach_hint);
  // supertile candidate lists of the two-level kernel (stream-ordered caching allocator)
  void *scratch = nullptr;
  if (int rc = psh_malloc(&scratch, psh::idw_scratch_bytes(m, n))) return rc;
  a.scratch = scratch;
  const hipError_t le = psh::launch_idw(a, c.stream);
  (void)psh_free(scratch);
  if (le != hipSuccess) return psh::fail(PSH_EHIP, "idw launch failed: %s", hipGetErrorString(le));
  return PSH_OK;
}

// Interpolation onto the unit pixel grid of an (m, n) image with the sample list, its length and
// the interpolator preamble in device memory (IdwDyn, written by vectors_finish): nothing about
// the samples is known on the host, the call only queues kernels.  k <= 0: every sample.
namespace psh {
int idw_resident(const float *xy_dev, const float *values_dev, int capacity, const IdwDyn *dyn_dev, int m, int n,
                 int k, double power, double dist_offset, float *out_dev, float *out_uv_dev) {
  if (!xy_dev || !values_dev || !dyn_dev || !out_dev) return fail(PSH_EINVAL, "idw: NULL pointer");
  if (capacity <= 0 || m <= 0 || n <= 0) return fail(PSH_EINVAL, "idw: invalid shape");
  if (k > 32) return fail(PSH_EUNSUPPORTED, "idw: k=%d > 32 is not implemented", k);
  if (!(power > 0.0)) return fail(PSH_EINVAL, "idw: power must be positive");
  Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  IdwArgs a;
  a.xy = xy_dev;
  a.uv = values_dev;
  a.out = out_dev;
  a.out_uv = out_uv_dev;
  a.L = capacity;
  a.k = k <= 0 ? 0x7fffffff : k;
  a.m = m;
  a.n = n;
  a.x0 = 0.f;
  a.dx = 1.f;
  a.y0 = 0.f;
  a.dy = 1.f;
  a.inv_res = 1.f;
  a.power = static_cast<float>(power);
  a.offset = static_cast<float>(dist_offset);
  a.dmax = 1.f;
  a.dyn = dyn_dev;
  void *scratch = nullptr;
  if (int rc = psh_malloc(&scratch, idw_scratch_bytes(m, n))) return rc;
  a.scratch = scratch;
  const hipError_t le = launch_idw(a, c.stream);
  (void)psh_free(scratch);
  if (le != hipSuccess) return fail(PSH_EHIP, "idw launch failed: %s", hipGetErrorString(le));
  return PSH_OK;
}
}  // namespace psh

extern "C" int psh_idw_host(const double *xy, const double *values, int L, int m, int n, double x0,
                            double dx, double y0, double dy, int k, double power,
                            double dist_offset, double *out) {
  PSH_REQUIRE_INIT();
  if (int rc = idw_check(L, k, m, n, dx, dy, power)) return rc;
  if (!xy || !values || !out) return psh::fail(PSH_EINVAL, "idw: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * n;
  std::vector<float> h_xy(2 * static_cast<size_t>(L)), h_uv(2 * static_cast<size_t>(L));
  // farthest any sample can be from any grid node: diagonal of the joint bounding box
  double xmin = fmin(x0, x0 + dx * (n - 1)), xmax = fmax(x0, x0 + dx * (n - 1));
  double ymin = fmin(y0, y0 + dy * (m - 1)), ymax = fmax(y0, y0 + dy * (m - 1));
  for (int i = 0; i < L; ++i) {
    h_xy[2 * i] = static_cast<float>(xy[2 * i]);
    h_xy[2 * i + 1] = static_cast<float>(xy[2 * i + 1]);
    h_uv[2 * i] = static_cast<float>(values[2 * i]);
    h_uv[2 * i + 1] = static_cast<float>(values[2 * i + 1]);
    xmin = fmin(xmin, xy[2 * i]);
    xmax = fmax(xmax, xy[2 * i]);
    ymin = fmin(ymin, xy[2 * i + 1]);
    ymax = fmax(ymax, xy[2 * i + 1]);
  }
  const double reach = hypot(xmax - xmin, ymax - ymin) * 1.001 + 1.0;
  float *d_xy = nullptr, *d_uv = nullptr, *d_out = nullptr;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(c.stream);
    if (d_xy) (void)psh_free(d_xy);
    if (d_uv) (void)psh_free(d_uv);
    if (d_out) (void)psh_free(d_out);
  };
#define PSH_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      cleanup();                                                                       \
      return psh::fail(_e == hipErrorOutOfMemory ? PSH_ENOMEM : PSH_EHIP, "%s failed: %s", #expr, \
                       hipGetErrorString(_e));                                         \
    }                                                                                  \
  } while (0)
  {
    void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
    int arc = psh_malloc(&p0, h_xy.size() * sizeof(float));
    if (!arc) arc = psh_malloc(&p1, h_uv.size() * sizeof(float));
    if (!arc) arc = psh_malloc(&p2, 2 * plane * sizeof(float));
    d_xy = static_cast<float *>(p0);
    d_uv = static_cast<float *>(p1);
    d_out = static_cast<float *>(p2);
    if (arc) {
      cleanup();
      return arc;
    }
  }
  PSH_TRY(hipMemcpyAsync(d_xy, h_xy.data(), h_xy.size() * sizeof(float), hipMemcpyHostToDevice, c.stream));
  PSH_TRY(hipMemcpyAsync(d_uv, h_uv.data(), h_uv.size() * sizeof(float), hipMemcpyHostToDevice, c.stream));
  int rc = psh_idw_dev(d_xy, d_uv, L, m, n, x0, dx, y0, dy, k, power, dist_offset, reach, d_out);
  if (rc != PSH_OK) {
    cleanup();
    return rc;
  }
  std::vector<float> h_out(2 * plane);
  PSH_TRY(hipMemcpyAsync(h_out.data(), d_out, 2 * plane * sizeof(float), hipMemcpyDeviceToHost, c.stream));
  PSH_TRY(hipStreamSynchronize(c.stream));
#undef PSH_TRY
  cleanup();
  for (size_t i = 0; i < 2 * plane; ++i) out[i] = static_cast<double>(h_out[i]);
  return PSH_OK;
}
