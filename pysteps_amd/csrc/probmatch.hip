// Empirical-CDF probability matching on the device:
// pysteps/postprocessing/probmatching.py:55-140, nonparam_match_empirical_cdf(initial, target) with
// ignore_indices=None, as the member loops call it (nowcasts/steps.py:1199, sprog.py:421,
// sseps.py:783,804).
//
// The reference argsorts both arrays (two O(N log N) host sorts of the whole grid).  Only two things
// are needed from them: the sorted values of the target and the rank of every pixel of the initial
// array - and in both arrays the "zeros" (pixels at the minimum, NaNs of the target) form one block of
// equal keys at the front whose internal order never reaches the output (:127-128).  So both arrays
// are ranked by one bucket pass over their wet values only:
//
//   stats      min / max / NaN / inf counts                     -> zero value z, bucket scale
//   bucket     b = floor((v - z) * 2^20 / (max - z)), monotone in v
//   target     (once per plan) counted per workgroup in an LDS table first - one global atomic per
//              workgroup and bucket; observations are quantised, thousands of pixels per distinct value -,
//              bucket starts by three small scan kernels, values scattered to their bucket's segment,
//              position inside the segment by counting the smaller values of the same bucket
//   initial    (every call) two partition passes over the bucket bits, 2^9 coarse x 2^11 fine, with the
//              counting in LDS and private output ranges from a scan - no device-scope atomic per
//              pixel (pm2_* below) -, then the same counting inside the bucket
//
// target: sorted wet values tw.  initial: rank r of every wet pixel, ties in pixel order (what
// argsort(kind="stable") gives; the reference's default quicksort leaves the order of tied WET
// values unspecified, see oracle/probmatch.py), and the output is written straight from it:
//   R = zeros_initial + r,  out = R < zeros_target ? z_target : tw[R - zeros_target],
// with the wet-area adjustment of :105-108 applied on the fly (out < p -> z_target; replacing the
// values below a threshold by the minimum keeps the sorted order).  p is np.percentile's "linear"
// method evaluated by one thread with the operation order of numpy (no contraction).
//
// Not handled here (PSH_EUNSUPPORTED, the Python shim hands those calls to the reference): more than
// 16384 values in one bucket that are not all equal (pathological spread), more than 16384 tied wet
// values in the initial array, infinities in the target.
#include <algorithm>

#include "common.h"

namespace psh {
namespace {

constexpr unsigned kBins = 1u << 20;
constexpr unsigned kScanBlock = 1024;  // buckets per workgroup of the scan kernels
constexpr unsigned kScanBlocks = kBins / kScanBlock;
constexpr unsigned kSmallBin = 256;    // up to here one thread ranks its value alone
constexpr unsigned kLargeLimit = 16384;
constexpr int kThreads = 256;
constexpr int kGrid = 2048;
constexpr int kLoads = 4;  // independent loads in flight per thread of the streaming kernels
constexpr unsigned kHashBits = 12, kHashSlots = 1u << kHashBits, kProbes = 8;  // per-workgroup bucket table (LDS)

enum { kStOk = 0, kStAllNan = 1, kStNonFinite = 2, kStTarget = 3, kStTies = 4 };

// index 0: initial array, 1: target array
struct PmHeader {
  unsigned int n_nan[2], n_inf[2];
  unsigned int wet[2], n_large[2], max_bin[2];
  int status, adjust;
  double z[2], scale[2], p;
};

struct PmPartial {  // statistics of one workgroup's share of an array
  double mn, mx;    // over the values that are not NaN (+inf / -inf if there is none)
  unsigned n_nan, n_inf;
};

__device__ __forceinline__ unsigned bin_of(double v, double z, double scale) {
#pragma clang fp contract(off)
  const double f = (v - z) * scale;
  return f >= static_cast<double>(kBins - 1) ? kBins - 1 : static_cast<unsigned>(f);
}

__device__ __forceinline__ void pm_init(PmHeader *h) {
  for (int y = 0; y < 2; ++y) {
    h->n_nan[y] = h->n_inf[y] = h->wet[y] = h->n_large[y] = h->max_bin[y] = 0u;
    h->z[y] = h->scale[y] = 0.0;
  }
  h->status = kStOk;
  h->adjust = 0;
  h->p = 0.0;
}

// 64-bit key of a double whose unsigned order is the doubles' order (steps_loop.hip min_key / from_min_key)
__device__ __forceinline__ double pm_from_min_key(unsigned long long k) {
  if (k == 0ull) return __longlong_as_double(0x7ff8000000000000ll);  // NaN
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double(static_cast<long long>(b));
}

__global__ __launch_bounds__(kThreads) void pm_stats(const double *__restrict__ a0, const double *__restrict__ a1,
                                                     size_t n, PmPartial *__restrict__ part) {
  const int y = blockIdx.y;
  const double *a = y ? a1 : a0;
  double mn = INFINITY, mx = -INFINITY;
  unsigned nn = 0, ni = 0;
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += kLoads * stride) {
    double v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) v[k] = i + k * stride < n ? a[i + k * stride] : a[i];  // a repeat changes nothing below
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const bool fresh = k == 0 || i + k * stride < n;
      const bool nan = v[k] != v[k];
      nn += (fresh && nan) ? 1u : 0u;
      ni += (fresh && (v[k] == INFINITY || v[k] == -INFINITY)) ? 1u : 0u;
      mn = v[k] < mn ? v[k] : mn;  // false for NaN
      mx = v[k] > mx ? v[k] : mx;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const double omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
    nn += __shfl_xor(nn, d);
    ni += __shfl_xor(ni, d);
  }
  __shared__ PmPartial s_part[kThreads / 64];
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = PmPartial{mn, mx, nn, ni};
  __syncthreads();
  if (threadIdx.x == 0) {
    PmPartial r = s_part[0];
    for (int w = 1; w < kThreads / 64; ++w) {
      r.mn = s_part[w].mn < r.mn ? s_part[w].mn : r.mn;
      r.mx = s_part[w].mx > r.mx ? s_part[w].mx : r.mx;
      r.n_nan += s_part[w].n_nan;
      r.n_inf += s_part[w].n_inf;
    }
    part[static_cast<size_t>(y) * gridDim.x + blockIdx.x] = r;  // no atomics: thousands on four addresses were the bound
  }
}

// steps.py:1221-1240 (the precipitation mask of the member loop, psh_steps_mask_dev's arithmetic) applied to the
// initial array IN PLACE, and the statistics of the masked values in the same sweep: the matching's first pass
// over the array is the mask's own
__global__ __launch_bounds__(kThreads) void pm_mask_stats(double *__restrict__ a, size_t n, const double *__restrict__ grey,
                                                          const unsigned char *__restrict__ keep,
                                                          const unsigned long long *__restrict__ min_key_in,
                                                          PmPartial *__restrict__ part) {
#pragma clang fp contract(off)
  const double base = pm_from_min_key(*min_key_in);
  double mn = INFINITY, mx = -INFINITY;
  unsigned nn = 0, ni = 0;
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += kLoads * stride) {
    double v[kLoads], g[kLoads];
    unsigned char kp[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const size_t j = i + k * stride;
      v[k] = j < n ? a[j] : 0.0;
      g[k] = (grey && j < n) ? grey[j] : 0.0;
      kp[k] = (keep && j < n) ? keep[j] : 0;
    }
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const size_t j = i + k * stride;
      if (j >= n) continue;
      double w = v[k];
      if (grey) {
        const double d = w - base;
        const double s = d * g[k];
        w = base + s;
        if (!(w > base)) w = base;
      } else if (!kp[k]) {
        w = base;
      }
      a[j] = w;
      const bool nan = w != w;
      nn += nan ? 1u : 0u;
      ni += (w == INFINITY || w == -INFINITY) ? 1u : 0u;
      mn = w < mn ? w : mn;  // false for NaN
      mx = w > mx ? w : mx;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const double omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
    nn += __shfl_xor(nn, d);
    ni += __shfl_xor(ni, d);
  }
  __shared__ PmPartial s_part[kThreads / 64];
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = PmPartial{mn, mx, nn, ni};
  __syncthreads();
  if (threadIdx.x == 0) {
    PmPartial r = s_part[0];
    for (int w = 1; w < kThreads / 64; ++w) {
      r.mn = s_part[w].mn < r.mn ? s_part[w].mn : r.mn;
      r.mx = s_part[w].mx > r.mx ? s_part[w].mx : r.mx;
      r.n_nan += s_part[w].n_nan;
      r.n_inf += s_part[w].n_inf;
    }
    part[blockIdx.x] = r;
  }
}

// one workgroup: finishes the statistics and fixes the bucket mapping.  `only` < 0: both arrays;
// 1: the target alone (a plan is being made: nothing is known about an initial array yet);
// 0: the initial array alone, the target's side of the header comes from `plan`
__global__ __launch_bounds__(kThreads) void pm_prepare(PmHeader *h, size_t n, const PmPartial *__restrict__ part,
                                                       int nparts, int only, const PmHeader *__restrict__ plan) {
  __shared__ PmPartial s_part[kThreads / 64];
  if (threadIdx.x == 0) pm_init(h);  // (thread 0 is the header's only writer in this kernel)
  for (int y = 0; y < 2; ++y) {
    if (only >= 0 && y != only) continue;  // uniform
    double mn = INFINITY, mx = -INFINITY;
    unsigned nn = 0, ni = 0;
    for (int i = threadIdx.x; i < nparts; i += kThreads) {
      const PmPartial r = part[static_cast<size_t>(y) * nparts + i];
      mn = r.mn < mn ? r.mn : mn;
      mx = r.mx > mx ? r.mx : mx;
      nn += r.n_nan;
      ni += r.n_inf;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const double omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
      mn = omn < mn ? omn : mn;
      mx = omx > mx ? omx : mx;
      nn += __shfl_xor(nn, d);
      ni += __shfl_xor(ni, d);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = PmPartial{mn, mx, nn, ni};
    __syncthreads();
    if (threadIdx.x == 0) {
      PmPartial r = s_part[0];
      for (int w = 1; w < kThreads / 64; ++w) {
        r.mn = s_part[w].mn < r.mn ? s_part[w].mn : r.mn;
        r.mx = s_part[w].mx > r.mx ? s_part[w].mx : r.mx;
        r.n_nan += s_part[w].n_nan;
        r.n_inf += s_part[w].n_inf;
      }
      h->n_nan[y] = r.n_nan;
      h->n_inf[y] = r.n_inf;
      h->z[y] = r.mn;  // :91, :101 (np.nanmin); +inf if there is nothing but NaNs
      const double range = r.mx - r.mn;  // may overflow to inf: scale 0, everything in bucket 0
      h->scale[y] = r.mx > r.mn ? static_cast<double>(kBins) / range : 0.0;
    }
  }
  if (threadIdx.x != 0) return;
  if (only == 0) {  // the target's statistics, wet count and verdict were fixed when the plan was made
    h->n_nan[1] = plan->n_nan[1];
    h->n_inf[1] = plan->n_inf[1];
    h->z[1] = plan->z[1];
    h->scale[1] = plan->scale[1];
    h->wet[1] = plan->wet[1];
  }
  // :81-82 (only NaNs), :93-96 (any non-finite value left: without ignore_indices every NaN / inf)
  if (only != 1 && h->n_nan[0] == n) {
    h->status = kStAllNan;
  } else if (only != 1 && h->n_nan[0] + h->n_inf[0] > 0) {
    h->status = kStNonFinite;
  } else if (only == 0 && plan->status != kStOk) {
    h->status = plan->status;  // kStTarget, kStTies
  } else if (h->n_nan[1] == n || h->n_inf[1] > 0) {
    h->status = kStTarget;
  }
}

// Bucket counters of one workgroup in LDS: an open-addressing table bucket -> count over everything the
// workgroup reads, flushed with ONE global atomic per occupied slot.  Observations are quantised
// (thousands of pixels per distinct value): their per-pixel atomics on a few hundred addresses were the
// bound.  Tags are never removed, so a bucket resolves to the same slot (or to "no room": the first
// kProbes positions taken by other buckets) every time it is looked up; buckets without room go to
// the global table one by one.
struct PmHash {
  unsigned tag[kHashSlots];  // bucket + 1, 0: free
  unsigned cnt[kHashSlots];
  unsigned base[kHashSlots];
};

__device__ __forceinline__ unsigned hash_home(unsigned bin) { return (bin * 2654435761u) >> (32 - kHashBits); }

__device__ __forceinline__ int hash_insert(PmHash &t, unsigned bin) {
  const unsigned home = hash_home(bin);
#pragma unroll 1
  for (unsigned p = 0; p < kProbes; ++p) {
    const unsigned s = (home + p) & (kHashSlots - 1);
    const unsigned old = atomicCAS(&t.tag[s], 0u, bin + 1u);
    if (old == 0u || old == bin + 1u) return static_cast<int>(s);
  }
  return -1;
}

__device__ __forceinline__ int hash_find(const PmHash &t, unsigned bin) {
  const unsigned home = hash_home(bin);
#pragma unroll 1
  for (unsigned p = 0; p < kProbes; ++p) {
    const unsigned s = (home + p) & (kHashSlots - 1);
    if (t.tag[s] == bin + 1u) return static_cast<int>(s);
  }
  return -1;
}

__device__ __forceinline__ void hash_clear(PmHash &t) {
  for (unsigned s = threadIdx.x; s < kHashSlots; s += kThreads) {
    t.tag[s] = 0u;
    t.cnt[s] = 0u;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void pm_hist_target(const double *__restrict__ a, size_t n,
                                                           const PmHeader *__restrict__ h, unsigned *table) {
  __shared__ PmHash t;
  if (h->status != kStOk) return;
  const double z = h->z[1], scale = h->scale[1];
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  hash_clear(t);
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += kLoads * stride) {
    double v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) v[k] = i + k * stride < n ? a[i + k * stride] : z;
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      if (v[k] > z) {  // false for NaN
        const unsigned b = bin_of(v[k], z, scale);
        const int s = hash_insert(t, b);
        if (s >= 0) {
          atomicAdd(&t.cnt[s], 1u);
        } else {
          atomicAdd(&table[b], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (unsigned s = threadIdx.x; s < kHashSlots; s += kThreads) {
    const unsigned c = t.cnt[s];
    if (c) atomicAdd(&table[t.tag[s] - 1u], c);
  }
}

__device__ __forceinline__ unsigned wave_incl_scan(unsigned v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// inclusive scan over the workgroup (NW waves); *total = sum over the workgroup
template <int NW>
__device__ __forceinline__ unsigned block_incl_scan(unsigned v, unsigned *s_wave, unsigned *total) {
  const int w = threadIdx.x >> 6;
  unsigned incl = wave_incl_scan(v);
  if ((threadIdx.x & 63) == 63) s_wave[w] = incl;
  __syncthreads();
  unsigned add = 0, all = 0;
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const unsigned s = s_wave[k];
    add += k < w ? s : 0u;
    all += s;
  }
  __syncthreads();
  *total = all;
  return incl + add;
}

// (y0: first array the launch works on - 1: the target alone, whose tables are the second halves)
__global__ __launch_bounds__(kThreads) void pm_bin_sums(const unsigned *__restrict__ count, unsigned *sums, int y0) {
  __shared__ unsigned s_wave[kThreads / 64];
  const int y = blockIdx.y + y0;
  const size_t base = static_cast<size_t>(y) * kBins + static_cast<size_t>(blockIdx.x) * kScanBlock;
  const uint4 c = reinterpret_cast<const uint4 *>(count + base)[threadIdx.x];
  unsigned total;
  (void)block_incl_scan<kThreads / 64>(c.x + c.y + c.z + c.w, s_wave, &total);
  if (threadIdx.x == 0) sums[y * kScanBlocks + blockIdx.x] = total;
}

__global__ __launch_bounds__(kScanBlocks) void pm_scan_sums(const unsigned *__restrict__ sums, unsigned *offs,
                                                            PmHeader *h, int y0) {
  __shared__ unsigned s_wave[kScanBlocks / 64];
  const int y = blockIdx.x + y0;
  const unsigned v = sums[y * kScanBlocks + threadIdx.x];
  unsigned total;
  const unsigned incl = block_incl_scan<kScanBlocks / 64>(v, s_wave, &total);
  offs[y * kScanBlocks + threadIdx.x] = incl - v;
  if (threadIdx.x == 0) h->wet[y] = total;
}

__global__ __launch_bounds__(kThreads) void pm_bin_starts(const unsigned *__restrict__ count,
                                                          const unsigned *__restrict__ offs, unsigned *start,
                                                          unsigned *cursor, unsigned *large, unsigned large_cap,
                                                          PmHeader *h, int y0) {
  __shared__ unsigned s_wave[kThreads / 64];
  const int y = blockIdx.y + y0;
  const size_t base = static_cast<size_t>(y) * kBins + static_cast<size_t>(blockIdx.x) * kScanBlock;
  const uint4 c = reinterpret_cast<const uint4 *>(count + base)[threadIdx.x];
  const unsigned mine = c.x + c.y + c.z + c.w;
  unsigned total;
  const unsigned incl = block_incl_scan<kThreads / 64>(mine, s_wave, &total);
  uint4 s;
  s.x = offs[y * kScanBlocks + blockIdx.x] + incl - mine;
  s.y = s.x + c.x;
  s.z = s.y + c.y;
  s.w = s.z + c.z;
  reinterpret_cast<uint4 *>(start + base)[threadIdx.x] = s;
  reinterpret_cast<uint4 *>(cursor + base)[threadIdx.x] = s;
  const unsigned cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (cs[k] > kSmallBin) {
      const unsigned at = atomicAdd(&h->n_large[y], 1u);
      if (at < large_cap)
        large[static_cast<size_t>(y) * large_cap + at] = blockIdx.x * kScanBlock + threadIdx.x * 4 + k;
      atomicMax(&h->max_bin[y], cs[k]);
    }
  }
}

// the target: pass A counts the workgroup's values per bucket in the LDS table, the flush reserves one
// range per occupied slot in the bucket's segment, pass B reads the values again and hands the
// positions out
__global__ __launch_bounds__(kThreads) void pm_scatter_target(const double *__restrict__ a, size_t n,
                                                              const PmHeader *__restrict__ h, unsigned *table,
                                                              double *__restrict__ sval) {
  __shared__ PmHash t;
  if (h->status != kStOk) return;
  const double z = h->z[1], scale = h->scale[1];
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  const size_t first = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x;
  hash_clear(t);
  for (size_t i = first; i < n; i += kLoads * stride) {
    double v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) v[k] = i + k * stride < n ? a[i + k * stride] : z;
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      if (v[k] > z) {
        const int s = hash_insert(t, bin_of(v[k], z, scale));
        if (s >= 0) atomicAdd(&t.cnt[s], 1u);
      }
    }
  }
  __syncthreads();
  for (unsigned s = threadIdx.x; s < kHashSlots; s += kThreads) {
    const unsigned c = t.cnt[s];
    if (c) {
      t.base[s] = atomicAdd(&table[t.tag[s] - 1u], c);
      t.cnt[s] = 0u;
    }
  }
  __syncthreads();
  for (size_t i = first; i < n; i += kLoads * stride) {
    double v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) v[k] = i + k * stride < n ? a[i + k * stride] : z;
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      if (v[k] > z) {
        const unsigned b = bin_of(v[k], z, scale);
        const int s = hash_find(t, b);
        const unsigned slot = s >= 0 ? t.base[s] + atomicAdd(&t.cnt[s], 1u) : atomicAdd(&table[b], 1u);
        sval[slot] = v[k];
      }
    }
  }
}

// output of one wet pixel of the initial array from its rank among the wet pixels
__device__ __forceinline__ void pm_emit(const PmHeader *__restrict__ h, size_t n, const double *__restrict__ tw,
                                        double *__restrict__ out, unsigned pixel, unsigned rank_wet) {
  const size_t r = (n - h->wet[0]) + rank_wet, zeros_trg = n - h->wet[1];
  PSH_DASSERT(pixel < n && rank_wet < h->wet[0] && r < n);  // a rank outside the wet block / a pixel outside the grid
  double val = r < zeros_trg ? h->z[1] : tw[r - zeros_trg];
  if (h->adjust && val < h->p) val = h->z[1];  // :108
  out[pixel] = val;
}

// the target: position inside the bucket = number of smaller values in it -> sorted wet values tw
// (one thread per value; the buckets above kSmallBin values by one workgroup each)
__global__ __launch_bounds__(kThreads) void pm_rank_small(const PmHeader *__restrict__ h,
                                                          const unsigned *__restrict__ count,
                                                          const unsigned *__restrict__ start,
                                                          const double *__restrict__ sval, double *__restrict__ tw) {
  if (h->status != kStOk) return;
  const double z = h->z[1], scale = h->scale[1];
  const unsigned wet = h->wet[1];
  count += kBins;
  start += kBins;
  const unsigned stride = gridDim.x * kThreads;
  for (unsigned s = blockIdx.x * kThreads + threadIdx.x; s < wet; s += stride) {
    const double v = sval[s];
    const unsigned b = bin_of(v, z, scale), c = count[b];
    if (c > kSmallBin) continue;  // pm_rank_large
    const unsigned st = start[b];
    unsigned less = 0;
    for (unsigned j = st; j < st + c; ++j) {
      const double vj = sval[j];
      less += (vj < v || (vj == v && j < s)) ? 1u : 0u;
    }
    tw[st + less] = v;
  }
}

__global__ __launch_bounds__(kThreads) void pm_rank_large(PmHeader *h, const unsigned *__restrict__ count,
                                                          const unsigned *__restrict__ start,
                                                          const unsigned *__restrict__ large, unsigned large_cap,
                                                          const double *__restrict__ sval, double *__restrict__ tw) {
  __shared__ double s_mn[kThreads / 64], s_mx[kThreads / 64];
  if (h->status != kStOk) return;
  count += kBins;
  start += kBins;
  large += large_cap;
  const unsigned n_large = h->n_large[1] < large_cap ? h->n_large[1] : large_cap;
  for (unsigned li = blockIdx.x; li < n_large; li += gridDim.x) {
    const unsigned b = large[li], c = count[b], st = start[b];
    double mn = INFINITY, mx = -INFINITY;
    for (unsigned j = threadIdx.x; j < c; j += kThreads) {
      const double v = sval[st + j];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const double omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
      mn = omn < mn ? omn : mn;
      mx = omx > mx ? omx : mx;
    }
    __syncthreads();  // the previous bucket's readers of s_mn / s_mx are done
    if ((threadIdx.x & 63) == 0) {
      s_mn[threadIdx.x >> 6] = mn;
      s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    mn = s_mn[0];
    mx = s_mx[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) {
      mn = s_mn[w] < mn ? s_mn[w] : mn;
      mx = s_mx[w] > mx ? s_mx[w] : mx;
    }
    if (mn == mx) {  // equal values (quantised observations): already sorted
      for (unsigned j = threadIdx.x; j < c; j += kThreads) tw[st + j] = mn;
      continue;
    }
    if (c > kLargeLimit) {
      if (threadIdx.x == 0) atomicExch(&h->status, kStTies);
      continue;
    }
    for (unsigned e = threadIdx.x; e < c; e += kThreads) {
      const double v = sval[st + e];
      unsigned less = 0;
      for (unsigned j = st; j < st + c; ++j) {
        const double vj = sval[j];
        less += (vj < v || (vj == v && j < st + e)) ? 1u : 0u;
      }
      tw[st + less] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The initial array without device-scope atomics (round 4).
// pm_hist_initial / pm_scatter_initial cost one global atomic per wet pixel each (8.4 M per call at
// 4096^2: 0.43 of the 1.15 ms), because a continuous forecast has a bucket of its own for nearly every
// value.  The 2^20 buckets are an MSD radix key: kCoarse x kFine.  Two partition passes move the
// atomics into LDS:
//   pm2_count    a workgroup owns kPxBlock consecutive PIXELS: LDS histogram over the coarse
//                buckets -> H[coarse][block]
//   pm2_sums / pm2_scan_sums / pm2_offsets   exclusive scan of H in (coarse, block) order: every
//                block gets a private range inside every coarse segment; wet count
//   pm2_scatter  the same pixels again: position = LDS cursor of the coarse bucket -> (value, pixel)
//                records grouped by coarse bucket; the dry pixels get their output here (:127-128)
//   pm2_refine   one workgroup per coarse segment, streaming it from memory: LDS histogram over its kFine
//                fine buckets, scan, second copy grouped by the full bucket, then the rank of every record
//                = bucket start + number of smaller (value, pixel) pairs in its bucket -> output
//   pm2_rank_large  the same for the listed buckets above kSmallBin values, one workgroup each
// Any distribution is handled (a segment is streamed, not staged: one outlier that squeezes a third
// of the values into one coarse bucket makes that workgroup slow, not wrong); the limits on TIED or
// bucket-sharing values are those of pm_rank_large.
constexpr unsigned kCoarseBits = 9, kCoarse = 1u << kCoarseBits;
constexpr unsigned kFineBits = 12, kFine = 1u << kFineBits;
constexpr unsigned kBinsI = kCoarse * kFine;  // 2^21 buckets for the initial array: half the crowding of 2^20
constexpr unsigned kPxBlock = 8192;  // pixels per workgroup of pm2_count / pm2_scatter
constexpr unsigned kScanChunk = 4096;  // entries of H per workgroup of the scan kernels
constexpr int kRefineThreads = 1024;

// (value, pixel) of a wet pixel; `tag` is filled in by pm2_refine: position of the record inside its
// bucket (bits 8..) and the bucket's size - 1 (bits 0..7), or kCrowded for buckets above kSmallBin values -
// so that the ranking pass finds the bucket's members from the record alone
struct PmRec {
  double v;
  unsigned idx, tag;
};
constexpr unsigned kCrowded = 0xffffffffu;

// bucket of the initial array: the header's scale is for kBins buckets, kBinsI is a power of two times that
__device__ __forceinline__ unsigned bin_i(double v, double z, double scale) {
#pragma clang fp contract(off)
  const double f = (v - z) * (scale * static_cast<double>(kBinsI / kBins));
  return f >= static_cast<double>(kBinsI - 1) ? kBinsI - 1 : static_cast<unsigned>(f);
}

__global__ __launch_bounds__(kThreads) void pm2_count(const double *__restrict__ a, size_t n,
                                                      const PmHeader *__restrict__ h, unsigned *__restrict__ H,
                                                      unsigned nblk) {
  __shared__ unsigned s_hist[kCoarse];
  for (unsigned c = threadIdx.x; c < kCoarse; c += kThreads) s_hist[c] = 0u;
  __syncthreads();
  if (h->status == kStOk) {
    const double z = h->z[0], scale = h->scale[0];
    const size_t first = static_cast<size_t>(blockIdx.x) * kPxBlock + threadIdx.x;
    for (unsigned k0 = 0; k0 < kPxBlock / kThreads; k0 += kLoads) {
      double v[kLoads];
#pragma unroll
      for (int k = 0; k < kLoads; ++k) {
        const size_t i = first + static_cast<size_t>(k0 + k) * kThreads;
        v[k] = i < n ? a[i] : z;
      }
#pragma unroll
      for (int k = 0; k < kLoads; ++k)
        if (v[k] > z) atomicAdd(&s_hist[bin_i(v[k], z, scale) >> kFineBits], 1u);  // false for NaN
    }
  }
  __syncthreads();
  for (unsigned c = threadIdx.x; c < kCoarse; c += kThreads) H[static_cast<size_t>(c) * nblk + blockIdx.x] = s_hist[c];
}

// exclusive scan of H[0 .. nent) in three steps: sums of kScanChunk entries, scan of the sums (one
// workgroup), offsets written back in place
__global__ __launch_bounds__(kThreads) void pm2_sums(const unsigned *__restrict__ H, size_t nent, unsigned *__restrict__ sums) {
  __shared__ unsigned s_wave[kThreads / 64];
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanChunk;
  unsigned mine = 0;
  for (unsigned k = 0; k < kScanChunk / kThreads; ++k) {
    const size_t i = base + static_cast<size_t>(threadIdx.x) * (kScanChunk / kThreads) + k;
    mine += i < nent ? H[i] : 0u;
  }
  unsigned total;
  (void)block_incl_scan<kThreads / 64>(mine, s_wave, &total);
  if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// :104-108: the wet area of the target above that of the initial array -> p = np.percentile(target,
// 100 * (1 - war)), method "linear" (numpy/lib/_function_base_impl.py: virtual index (n - 1) * q,
// _get_indexes, _lerp), every operation in numpy's order and unfused
__device__ __forceinline__ void pm_threshold(PmHeader *h, size_t n, const double *__restrict__ tw, unsigned wi) {
#pragma clang fp contract(off)
  if (h->status != kStOk) return;
  const unsigned wt = h->wet[1];
  if (wt <= wi) return;
  const size_t zeros_trg = n - wt;
  const double war = static_cast<double>(wi) / static_cast<double>(n);
  const double one_minus = 1.0 - war;
  const double percent = 100.0 * one_minus;
  const double q = percent / 100.0;
  const double last = static_cast<double>(n - 1);
  const double virt = last * q;
  size_t ia, ib;
  double t;
  if (virt >= last) {  // _get_indexes: both -1, gamma = virt - (-1)
    ia = ib = n - 1;
    t = virt + 1.0;
  } else {
    const double fl = floor(virt);
    ia = static_cast<size_t>(fl);
    ib = ia + 1;
    t = virt - fl;
  }
  const double a = ia < zeros_trg ? h->z[1] : tw[ia - zeros_trg];
  const double b = ib < zeros_trg ? h->z[1] : tw[ib - zeros_trg];
  const double diff = b - a;
  double p = a + diff * t;
  if (t >= 0.5) p = b - diff * (1.0 - t);
  h->p = p;
  h->adjust = 1;
}

// offsets written back in place; every workgroup adds the sums in front of its own chunk itself (a few hundred
// words) - no scan kernel of the sums -, workgroup 0 also the total = wet count of the initial array, with which
// its first thread fixes the wet-area threshold (no kernel of its own either)
__global__ __launch_bounds__(kThreads) void pm2_offsets(unsigned *__restrict__ H, size_t nent,
                                                        const unsigned *__restrict__ sums, unsigned nsums, PmHeader *h,
                                                        size_t n, const double *__restrict__ tw) {
  __shared__ unsigned s_wave[kThreads / 64];
  __shared__ unsigned s_before, s_all;
  unsigned before = 0, all = 0;
  for (unsigned j = threadIdx.x; j < nsums; j += kThreads) {
    const unsigned v = sums[j];
    before += j < blockIdx.x ? v : 0u;
    all += v;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    before += __shfl_xor(before, d);
    all += __shfl_xor(all, d);
  }
  if (threadIdx.x == 0) s_before = s_all = 0u;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s_before, before);  // (integers: the order does not matter)
    atomicAdd(&s_all, all);
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    h->wet[0] = s_all;
    pm_threshold(h, n, tw, s_all);
  }
  constexpr unsigned kPer = kScanChunk / kThreads;
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanChunk + static_cast<size_t>(threadIdx.x) * kPer;
  unsigned v[kPer], mine = 0;
#pragma unroll
  for (unsigned k = 0; k < kPer; ++k) {
    v[k] = base + k < nent ? H[base + k] : 0u;
    mine += v[k];
  }
  unsigned total;
  const unsigned incl = block_incl_scan<kThreads / 64>(mine, s_wave, &total);
  unsigned run = s_before + incl - mine;
#pragma unroll
  for (unsigned k = 0; k < kPer; ++k) {
    if (base + k < nent) H[base + k] = run;
    run += v[k];
  }
}

__global__ __launch_bounds__(kThreads) void pm2_scatter(const double *__restrict__ a, size_t n,
                                                        const PmHeader *__restrict__ h, const unsigned *__restrict__ H,
                                                        unsigned nblk, PmRec *__restrict__ rec, double *__restrict__ out) {
  __shared__ unsigned s_cur[kCoarse];
  if (h->status != kStOk) return;
  for (unsigned c = threadIdx.x; c < kCoarse; c += kThreads) s_cur[c] = H[static_cast<size_t>(c) * nblk + blockIdx.x];
  __syncthreads();
  const double z = h->z[0], scale = h->scale[0], z_trg = h->z[1];
  const size_t first = static_cast<size_t>(blockIdx.x) * kPxBlock + threadIdx.x;
  for (unsigned k0 = 0; k0 < kPxBlock / kThreads; k0 += kLoads) {
    double v[kLoads];
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const size_t i = first + static_cast<size_t>(k0 + k) * kThreads;
      v[k] = i < n ? a[i] : z;
    }
#pragma unroll
    for (int k = 0; k < kLoads; ++k) {
      const size_t i = first + static_cast<size_t>(k0 + k) * kThreads;
      if (v[k] > z) {
        const unsigned pos = atomicAdd(&s_cur[bin_i(v[k], z, scale) >> kFineBits], 1u);
        rec[pos] = PmRec{v[k], static_cast<unsigned>(i), 0u};
      } else if (i < n) {
        out[i] = z_trg;
      }
    }
  }
}

__global__ __launch_bounds__(kRefineThreads) void pm2_refine(PmHeader *h, const unsigned *__restrict__ H, unsigned nblk,
                                                             const PmRec *__restrict__ rec_in, PmRec *__restrict__ rec_out,
                                                             uint2 *large, unsigned large_cap) {
  __shared__ unsigned s_cnt[kFine], s_start[kFine], s_cur[kFine];
  __shared__ unsigned s_wave[kRefineThreads / 64];
  if (h->status != kStOk) return;
  const unsigned c = blockIdx.x;
  const double z = h->z[0], scale = h->scale[0];
  const unsigned seg0 = H[static_cast<size_t>(c) * nblk];
  const unsigned seg1 = c + 1 < kCoarse ? H[static_cast<size_t>(c + 1) * nblk] : h->wet[0];
  for (unsigned f = threadIdx.x; f < kFine; f += kRefineThreads) {
    s_cnt[f] = 0u;
    s_cur[f] = 0u;
  }
  __syncthreads();
  constexpr unsigned kFly = 4;  // records in flight per thread (a workgroup streams its segment at memory latency)
  for (unsigned e0 = seg0 + threadIdx.x; e0 < seg1; e0 += kFly * kRefineThreads) {
    double v[kFly];
#pragma unroll
    for (unsigned k = 0; k < kFly; ++k) v[k] = e0 + k * kRefineThreads < seg1 ? rec_in[e0 + k * kRefineThreads].v : 0.0;
#pragma unroll
    for (unsigned k = 0; k < kFly; ++k)
      if (e0 + k * kRefineThreads < seg1) atomicAdd(&s_cnt[bin_i(v[k], z, scale) & (kFine - 1)], 1u);
  }
  __syncthreads();
  // exclusive scan of the fine counts (kPer per thread); the crowded buckets are listed with their range
  constexpr unsigned kPer = kFine / kRefineThreads;
  unsigned cs[kPer], mine = 0;
#pragma unroll
  for (unsigned k = 0; k < kPer; ++k) {
    cs[k] = s_cnt[kPer * threadIdx.x + k];
    mine += cs[k];
  }
  unsigned total;
  const unsigned incl = block_incl_scan<kRefineThreads / 64>(mine, s_wave, &total);
  unsigned run = seg0 + incl - mine;
#pragma unroll
  for (unsigned k = 0; k < kPer; ++k) {
    s_start[kPer * threadIdx.x + k] = run;
    if (cs[k] > kSmallBin) {
      const unsigned at = atomicAdd(&h->n_large[0], 1u);
      if (at < large_cap) large[at] = make_uint2(run, cs[k]);
      atomicMax(&h->max_bin[0], cs[k]);
    }
    run += cs[k];
  }
  __syncthreads();
  for (unsigned e0 = seg0 + threadIdx.x; e0 < seg1; e0 += kFly * kRefineThreads) {
    PmRec r[kFly];
#pragma unroll
    for (unsigned k = 0; k < kFly; ++k)
      r[k] = e0 + k * kRefineThreads < seg1 ? rec_in[e0 + k * kRefineThreads] : PmRec{0.0, 0u, 0u};
#pragma unroll
    for (unsigned k = 0; k < kFly; ++k) {
      if (e0 + k * kRefineThreads < seg1) {
        const unsigned f = bin_i(r[k].v, z, scale) & (kFine - 1);
        const unsigned at = atomicAdd(&s_cur[f], 1u), cnt = s_cnt[f];
        r[k].tag = cnt > kSmallBin ? kCrowded : ((at << 8) | (cnt - 1u));
        PSH_DASSERT(at < cnt && s_start[f] + at >= seg0 && s_start[f] + at < seg1);  // inside the fine bucket's segment
        rec_out[s_start[f] + at] = r[k];
      }
    }
  }
}

// position inside the bucket = number of smaller (value, pixel) pairs; the bucket's members are found
// from the record's tag.  Two records per thread side by side: the kernel is a chain of dependent loads
// (record -> bucket members), a second chain fills the waiting time of the first.
// The workgroups behind the first `small_blocks` take the crowded buckets (more than kSmallBin values), one bucket
// per workgroup and turn (what pm2_rank_large was as a launch of its own; as a rule there is none).
__global__ __launch_bounds__(kThreads) void pm2_rank(PmHeader *h, size_t n, const PmRec *__restrict__ rec,
                                                     const double *__restrict__ tw, double *__restrict__ out,
                                                     unsigned small_blocks, const uint2 *__restrict__ large,
                                                     unsigned large_cap) {
  if (h->status != kStOk) return;
  if (blockIdx.x >= small_blocks) {
    const unsigned n_large = h->n_large[0] < large_cap ? h->n_large[0] : large_cap;
    for (unsigned li = blockIdx.x - small_blocks; li < n_large; li += gridDim.x - small_blocks) {
      const unsigned st = large[li].x, c = large[li].y;
      if (c > kLargeLimit) {
        if (threadIdx.x == 0) atomicExch(&h->status, kStTies);
        continue;
      }
      for (unsigned e = threadIdx.x; e < c; e += kThreads) {
        const PmRec me = rec[st + e];
        unsigned less = 0;
        for (unsigned j = st; j < st + c; ++j) {
          const PmRec o = rec[j];
          less += (o.v < me.v || (o.v == me.v && o.idx < me.idx)) ? 1u : 0u;
        }
        pm_emit(h, n, tw, out, me.idx, st + less);
      }
    }
    return;
  }
  const unsigned wet = h->wet[0];
  const unsigned stride = small_blocks * kThreads;
  for (unsigned s0 = blockIdx.x * kThreads + threadIdx.x; s0 < wet; s0 += 2 * stride) {
    const unsigned s1 = s0 + stride;
    const bool two = s1 < wet;
    const PmRec me0 = rec[s0], me1 = rec[two ? s1 : s0];
    const unsigned n0 = me0.tag == kCrowded ? 0u : (me0.tag & 0xffu) + 1u;  // crowded buckets: pm2_rank_large
    const unsigned n1 = (!two || me1.tag == kCrowded) ? 0u : (me1.tag & 0xffu) + 1u;
    const unsigned st0 = s0 - (me0.tag >> 8), st1 = s1 - (me1.tag >> 8);
    unsigned less0 = 0, less1 = 0;
    const unsigned both = n0 < n1 ? n0 : n1;
    unsigned j = 0;
    for (; j < both; ++j) {
      const PmRec o0 = rec[st0 + j], o1 = rec[st1 + j];
      less0 += (o0.v < me0.v || (o0.v == me0.v && o0.idx < me0.idx)) ? 1u : 0u;
      less1 += (o1.v < me1.v || (o1.v == me1.v && o1.idx < me1.idx)) ? 1u : 0u;
    }
    for (unsigned q = j; q < n0; ++q) {
      const PmRec o = rec[st0 + q];
      less0 += (o.v < me0.v || (o.v == me0.v && o.idx < me0.idx)) ? 1u : 0u;
    }
    for (unsigned q = j; q < n1; ++q) {
      const PmRec o = rec[st1 + q];
      less1 += (o.v < me1.v || (o.v == me1.v && o.idx < me1.idx)) ? 1u : 0u;
    }
    if (n0) pm_emit(h, n, tw, out, me0.idx, st0 + less0);
    if (n1) pm_emit(h, n, tw, out, me1.idx, st1 + less1);
  }
}

}  // namespace
}  // namespace psh

static int probmatch_status_to_rc(int status) {
  using namespace psh;
  switch (status) {
    case kStOk:
      return PSH_OK;
    case kStAllNan:
      return fail(PSH_EINVAL, "Initial array contains only nans.");
    case kStNonFinite:
      return fail(PSH_EINVAL, "Initial array contains non-finite values outside ignore_indices mask.");
    case kStTarget:
      return fail(PSH_EUNSUPPORTED, "probmatch: target array without finite values or with infinities");
    default:
      return fail(PSH_EUNSUPPORTED, "probmatch: more than %u tied or bucket-sharing wet values", kLargeLimit);
  }
}

// A target that stays the same from call to call - the observation every member of a STEPS ensemble is
// matched against at every time step (nowcasts/steps.py:1199) - needs its half of the work once: the
// plan keeps its statistics, its verdict and its sorted wet values (36 % of a call at 4096^2).
struct PmPlan {
  size_t count;
  void *blk;  // device: [PmHeader | sorted wet values of the target, `count` doubles]
  const psh::PmHeader *header() const { return static_cast<const psh::PmHeader *>(blk); }
  const double *tw() const { return reinterpret_cast<const double *>(static_cast<const char *>(blk) + 256); }
};

// plan == nullptr, make == nullptr: one complete call.  make: only the target's half, kept in *make.
// plan: only the initial array's half against the plan.
struct PmMask {  // psh_steps_mask_probmatch_dev: exactly one of grey / keep
  const double *grey;
  const unsigned char *keep;
  const unsigned long long *min_key;
};

static int probmatch_run(const double *initial_dev, const double *target_dev, size_t count, double *out_dev,
                         int *status_dev, const PmPlan *plan = nullptr, PmPlan *make = nullptr,
                         const PmMask *mask = nullptr) {
  using namespace psh;
  PSH_REQUIRE_INIT();
  if ((!make && (!initial_dev || !out_dev)) || (!plan && !target_dev)) return fail(PSH_EINVAL, "probmatch: NULL pointer");
  if (count == 0) return fail(PSH_EINVAL, "probmatch: empty arrays");
  if (count > 0x7fffffffull) return fail(PSH_EUNSUPPORTED, "probmatch: more than 2^31-1 pixels");
  Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));

  // [header | statistics partials 2 x 2048 | counts 2 x B | starts 2 x B | cursors 2 x B | block sums 2 x 1024 | block offsets 2 x 1024 |
  //  large-bucket lists 2 x cap | scattered target values N | sorted wet target values N |
  //  H: coarse x pixel blocks | sums of H | records N (by coarse bucket) | records N (by bucket)]
  const unsigned large_cap = static_cast<unsigned>(count / kSmallBin + 1);
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  const size_t table_bytes = 2 * static_cast<size_t>(kBins) * sizeof(unsigned);
  const size_t off_part = 256, part_bytes = 2 * static_cast<size_t>(kGrid) * sizeof(PmPartial);
  const size_t off_count = off_part + part_bytes, off_start = off_count + table_bytes, off_cursor = off_start + table_bytes;
  const size_t off_sums = off_cursor + table_bytes, off_offs = off_sums + 2 * kScanBlocks * sizeof(unsigned);
  const size_t off_large = off_offs + 2 * kScanBlocks * sizeof(unsigned);
  const size_t off_sval = up(off_large + 3 * static_cast<size_t>(large_cap) * sizeof(unsigned));
  const size_t off_tw = up(off_sval + (plan ? 0 : count * sizeof(double)));
  const unsigned nblk = static_cast<unsigned>((count + kPxBlock - 1) / kPxBlock);
  const size_t nent = static_cast<size_t>(kCoarse) * nblk;
  const unsigned nsums = static_cast<unsigned>((nent + kScanChunk - 1) / kScanChunk);
  const size_t off_h = up(off_tw + (plan ? 0 : count * sizeof(double)));
  const size_t off_hsums = up(off_h + (make ? 0 : nent * sizeof(unsigned)));
  const size_t off_rec_a = up(off_hsums + (make ? 0 : static_cast<size_t>(nsums) * sizeof(unsigned)));
  const size_t off_rec_b = up(off_rec_a + (make ? 0 : count * sizeof(PmRec)));
  const size_t total = off_rec_b + (make ? 0 : count * sizeof(PmRec));
  static_assert(sizeof(PmHeader) <= 256, "header block");
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, total)) return rc;
  char *base = static_cast<char *>(blk);
  PmHeader *h = reinterpret_cast<PmHeader *>(base);
  PmPartial *part = reinterpret_cast<PmPartial *>(base + off_part);
  unsigned *cnt = reinterpret_cast<unsigned *>(base + off_count);
  unsigned *start = reinterpret_cast<unsigned *>(base + off_start);
  unsigned *cursor = reinterpret_cast<unsigned *>(base + off_cursor);
  unsigned *sums = reinterpret_cast<unsigned *>(base + off_sums);
  unsigned *offs = reinterpret_cast<unsigned *>(base + off_offs);
  // (crowded buckets - target: bucket numbers in the second half; initial: (start, count) pairs - 2 x large_cap words each)
  unsigned *large = reinterpret_cast<unsigned *>(base + off_large) + large_cap;
  uint2 *large2 = reinterpret_cast<uint2 *>(base + off_large);
  double *sval = reinterpret_cast<double *>(base + off_sval);
  double *tw = plan ? const_cast<double *>(plan->tw()) : reinterpret_cast<double *>(base + off_tw);
  unsigned *H = reinterpret_cast<unsigned *>(base + off_h);
  unsigned *hsums = reinterpret_cast<unsigned *>(base + off_hsums);
  PmRec *rec_a = reinterpret_cast<PmRec *>(base + off_rec_a);
  PmRec *rec_b = reinterpret_cast<PmRec *>(base + off_rec_b);

  int status = -1;
  auto run = [&]() -> int {
    // every workgroup of the streaming kernels reads at least kLoads x kThreads values
    const int grid = static_cast<int>(
        std::min<size_t>(kGrid, (count + kLoads * kThreads - 1) / (static_cast<size_t>(kLoads) * kThreads)));
    const int grid_large = 1024;
    const int halves = plan ? 1 : 2;
    hipStream_t s = c.stream;
    if (!plan) PSH_HIP(hipMemsetAsync(cnt + kBins, 0, table_bytes / 2, s));  // the target's bucket counts
    // (the header is reset by pm_prepare; nothing before it reads one)
    int nparts = grid;
    if (mask) {  // the member loop's precipitation mask in place + the statistics of what it leaves, one sweep
      nparts = kGrid;
      hipLaunchKernelGGL(pm_mask_stats, dim3(nparts), dim3(kThreads), 0, s, const_cast<double *>(initial_dev), count, mask->grey,
                         mask->keep, mask->min_key, part);
    } else {
      hipLaunchKernelGGL(pm_stats, dim3(grid, halves), dim3(kThreads), 0, s, make ? target_dev : initial_dev, target_dev, count,
                         part);
    }
    hipLaunchKernelGGL(pm_prepare, dim3(1), dim3(kThreads), 0, s, h, count, part, nparts, plan ? 0 : (make ? 1 : -1),
                       plan ? plan->header() : static_cast<const PmHeader *>(nullptr));
    if (!plan) {  // target: bucket counts, bucket starts, sorted wet values
      hipLaunchKernelGGL(pm_hist_target, dim3(grid), dim3(kThreads), 0, s, target_dev, count, h, cnt + kBins);
      hipLaunchKernelGGL(pm_bin_sums, dim3(kScanBlocks, 1), dim3(kThreads), 0, s, cnt, sums, 1);
      hipLaunchKernelGGL(pm_scan_sums, dim3(1), dim3(kScanBlocks), 0, s, sums, offs, h, 1);
      hipLaunchKernelGGL(pm_bin_starts, dim3(kScanBlocks, 1), dim3(kThreads), 0, s, cnt, offs, start, cursor, large,
                         large_cap, h, 1);
      hipLaunchKernelGGL(pm_scatter_target, dim3(grid), dim3(kThreads), 0, s, target_dev, count, h, cursor + kBins, sval);
      hipLaunchKernelGGL(pm_rank_small, dim3(grid), dim3(kThreads), 0, s, h, cnt, start, sval, tw);
      hipLaunchKernelGGL(pm_rank_large, dim3(grid_large), dim3(kThreads), 0, s, h, cnt, start, large, large_cap, sval, tw);
    }
    if (make) {  // keep the header and the sorted values
      PSH_HIP(hipMemcpyAsync(make->blk, h, sizeof(PmHeader), hipMemcpyDeviceToDevice, s));
      PSH_HIP(hipMemcpyAsync(static_cast<char *>(make->blk) + 256, tw, count * sizeof(double), hipMemcpyDeviceToDevice, s));
      PSH_HIP(hipGetLastError());
      status = kStOk;  // the verdict on the target travels in the plan
      return PSH_OK;
    }
    // initial: two partition passes (LDS atomics only), ranks inside the buckets -> output
    hipLaunchKernelGGL(pm2_count, dim3(nblk), dim3(kThreads), 0, s, initial_dev, count, h, H, nblk);
    hipLaunchKernelGGL(pm2_sums, dim3(nsums), dim3(kThreads), 0, s, H, nent, hsums);
    hipLaunchKernelGGL(pm2_offsets, dim3(nsums), dim3(kThreads), 0, s, H, nent, hsums, nsums, h, count, tw);
    hipLaunchKernelGGL(pm2_scatter, dim3(nblk), dim3(kThreads), 0, s, initial_dev, count, h, H, nblk, rec_a, out_dev);
    hipLaunchKernelGGL(pm2_refine, dim3(kCoarse), dim3(kRefineThreads), 0, s, h, H, nblk, rec_a, rec_b, large2, large_cap);
    hipLaunchKernelGGL(pm2_rank, dim3(grid + grid_large), dim3(kThreads), 0, s, h, count, rec_b, tw, out_dev,
                       static_cast<unsigned>(grid), large2, large_cap);
    PSH_HIP(hipGetLastError());
    if (status_dev) {  // the caller reads the status later (resident member loop: one wait per time step)
      PSH_HIP(hipMemcpyAsync(status_dev, &h->status, sizeof(int), hipMemcpyDeviceToDevice, s));
      status = kStOk;
      return PSH_OK;
    }
    static void *pinned = nullptr;
    if (int rc = persistent_pinned(&pinned, 64)) return rc;
    PSH_HIP(hipMemcpyAsync(pinned, &h->status, sizeof(int), hipMemcpyDeviceToHost, s));
    PSH_HIP(hipStreamSynchronize(s));
    status = *static_cast<const int *>(pinned);
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  if (rc) return rc;
  return probmatch_status_to_rc(status);
}

extern "C" int psh_probmatch_dev(const double *initial_dev, const double *target_dev, size_t count, double *out_dev) {
  return probmatch_run(initial_dev, target_dev, count, out_dev, nullptr);
}

extern "C" int psh_probmatch_async_dev(const double *initial_dev, const double *target_dev, size_t count, double *out_dev,
                                       int *status_dev) {
  if (!status_dev) return psh::fail(PSH_EINVAL, "probmatch_async: NULL status pointer");
  return probmatch_run(initial_dev, target_dev, count, out_dev, status_dev);
}

extern "C" int psh_probmatch_plan_create(const double *target_dev, size_t count, void **plan_out) {
  if (!plan_out) return psh::fail(PSH_EINVAL, "probmatch_plan_create: NULL pointer");
  *plan_out = nullptr;
  if (!target_dev || count == 0 || count > 0x7fffffffull) return psh::fail(PSH_EINVAL, "probmatch_plan_create: invalid target");
  PmPlan *plan = new PmPlan{count, nullptr};
  int rc = psh_malloc(&plan->blk, 256 + count * sizeof(double));
  if (rc == PSH_OK) rc = probmatch_run(nullptr, target_dev, count, nullptr, nullptr, nullptr, plan);
  if (rc != PSH_OK) {
    if (plan->blk) (void)psh_free(plan->blk);
    delete plan;
    return rc;
  }
  *plan_out = plan;
  return PSH_OK;
}

extern "C" int psh_probmatch_plan_destroy(void *plan_handle) {
  PmPlan *plan = static_cast<PmPlan *>(plan_handle);
  if (!plan) return PSH_OK;
  const int rc = plan->blk ? psh_free(plan->blk) : PSH_OK;  // stream-ordered: queued matchings still read it
  delete plan;
  return rc;
}

extern "C" int psh_probmatch_planned_dev(const void *plan_handle, const double *initial_dev, size_t count, double *out_dev,
                                         int *status_dev) {
  const PmPlan *plan = static_cast<const PmPlan *>(plan_handle);
  if (!plan || !plan->blk) return psh::fail(PSH_EINVAL, "probmatch_planned: NULL plan");
  if (count != plan->count) return psh::fail(PSH_EINVAL, "probmatch_planned: the plan was made for %zu values, not %zu", plan->count, count);
  return probmatch_run(initial_dev, nullptr, count, out_dev, status_dev, plan, nullptr);
}

// psh_steps_mask_dev(field, ...) followed by psh_probmatch_planned_dev(plan, field, ...) with the mask applied by the
// matching's own first sweep (field_dev holds the masked field afterwards, as after psh_steps_mask_dev)
extern "C" int psh_steps_mask_probmatch_dev(const void *plan_handle, double *field_dev, size_t count,
                                            const double *grey_mask_dev, const unsigned char *keep_mask_dev,
                                            const unsigned long long *min_key_dev, double *out_dev, int *status_dev) {
  const PmPlan *plan = static_cast<const PmPlan *>(plan_handle);
  if (!plan || !plan->blk) return psh::fail(PSH_EINVAL, "steps_mask_probmatch: NULL plan");
  if (count != plan->count)
    return psh::fail(PSH_EINVAL, "steps_mask_probmatch: the plan was made for %zu values, not %zu", plan->count, count);
  if (!field_dev || !min_key_dev || (!grey_mask_dev == !keep_mask_dev))
    return psh::fail(PSH_EINVAL, "steps_mask_probmatch: field, minimum and exactly one of the two masks are required");
  if (field_dev == out_dev) return psh::fail(PSH_EINVAL, "steps_mask_probmatch: the matched field needs an array of its own");
  const PmMask mask{grey_mask_dev, keep_mask_dev, min_key_dev};
  return probmatch_run(field_dev, nullptr, count, out_dev, status_dev, plan, nullptr, &mask);
}

// The k-th smallest value of a field without NaNs (0-based) - what compute_percentile_mask
// (pysteps/nowcasts/utils.py:102-138: a full sort, then ONE element of it) needs for the S-PROG mask of the
// STEPS member loop (steps.py:1113-1114).  The plan of the field as a matching target IS its sorted form:
// `zeros` copies of the minimum followed by the sorted larger values.
extern "C" int psh_order_statistic_dev(const double *field_dev, size_t count, size_t index, double *value_host) {
  using namespace psh;
  if (!value_host) return fail(PSH_EINVAL, "order_statistic: NULL pointer");
  if (index >= count) return fail(PSH_EINVAL, "order_statistic: index %zu outside 0..%zu", index, count - 1);
  void *handle = nullptr;
  if (int rc = psh_probmatch_plan_create(field_dev, count, &handle)) return rc;
  PmPlan *plan = static_cast<PmPlan *>(handle);
  Context &c = ctx();
  int rc = PSH_OK;
  {
    std::lock_guard<std::recursive_mutex> lock(c.mu);
    static void *pinned = nullptr;
    auto run = [&]() -> int {
      if (int r = persistent_pinned(&pinned, 512)) return r;
      PmHeader *h = static_cast<PmHeader *>(pinned);
      PSH_HIP(hipMemcpyAsync(h, plan->blk, sizeof(PmHeader), hipMemcpyDeviceToHost, c.stream));
      PSH_HIP(hipStreamSynchronize(c.stream));
      if (h->status != kStOk) return probmatch_status_to_rc(h->status);
      if (h->n_nan[1] != 0) return fail(PSH_EINVAL, "order_statistic: the field contains NaNs");
      const size_t zeros = count - h->wet[1];
      if (index < zeros) {
        *value_host = h->z[1];
        return PSH_OK;
      }
      double *slot = reinterpret_cast<double *>(static_cast<char *>(pinned) + 256);
      PSH_HIP(hipMemcpyAsync(slot, plan->tw() + (index - zeros), sizeof(double), hipMemcpyDeviceToHost, c.stream));
      PSH_HIP(hipStreamSynchronize(c.stream));
      *value_host = *slot;
      return PSH_OK;
    };
    rc = run();
  }
  (void)psh_probmatch_plan_destroy(handle);
  return rc;
}

extern "C" int psh_probmatch_status(int status) {
  return probmatch_status_to_rc(status);
}
