// The sparse stage of dense Lucas-Kanade without a host hand-over.
//
// Two single-workgroup kernels that keep the ordered, data-dependent parts of
// pysteps/motion/lucaskanade.py:207-274 on the device, so that one estimate is a chain of
// kernel launches with no device->host copy in the middle:
//
//  * corner_order  - the walk of cv2.goodFeaturesToTrack over the corner candidates
//    (pysteps/feature/shitomasi.py:152-166; OpenCV featureselect.cpp): strongest response first,
//    ties by the higher pixel address, a candidate is accepted unless an accepted corner of a
//    neighbouring min-distance cell lies closer than min_distance, stop at max_corners.
//    Only the head of the ordered list is ever walked, so nothing is sorted as a whole: a
//    1024-bin histogram over the key range picks the threshold key above which ~3 k candidates
//    lie, those are gathered into LDS, ordered there (bitonic network) and walked in batches of
//    64 - every wave tests the batch against its share of the accepted corners, wave 0 resolves
//    the conflicts inside the batch in order.  If the head runs out before max_corners corners
//    are accepted, the next chunk below the threshold is selected the same way (any number of
//    candidates, any response distribution: an overfull bin is refined 10 key bits at a time).
//  * vectors_finish - what follows the outlier test (lucaskanade.py:254-274): drop the flagged
//    vectors, decluster (pysteps/utils/cleansing.py:21-121: cell = floor(xy / scale), cells in
//    lexicographic order, component-wise medians), the trivial cases of the interpolator
//    (pysteps/decorators.py:199-208) and the float32 sample list + sample count the IDW kernels
//    read from device memory (IdwDyn).
//
// Both replace host code of earlier versions (a rocPRIM sort + host pass, psh_decluster_host);
// results are bit-identical to those (tests/test_lk_gpu.py compares against the host entry
// points and the oracle).
#include <cstddef>

#include "common.h"

namespace psh {
namespace {

using CornerKey = unsigned long long;

constexpr int kOrdThreads = 1024;
constexpr int kOrdWaves = kOrdThreads / 64;
constexpr int kOrdBins = 1024;
constexpr int kChunkCap = 4096;     // most candidates ordered at a time (32 KiB of keys in LDS); launch_corner_order picks <= this
constexpr int kMaxCornersDev = 2048;  // accepted corners kept in LDS
constexpr int kHashSlots = 8192;      // bucket (hash of the cell key) -> chain of accepted corners; mean chain length <= 1/4

// descending (DESC) or ascending bitonic sort of the first `p2` (power of two, >= 128) LDS entries;
// a thread owns whole compare-exchange pairs (p2 / 2 of them per step).  Steps with a partner
// distance j <= 64 stay inside the 128 consecutive entries a WAVE owns (threads t and t + 64 q map
// to the same 128-entry block), so they only need the wave's own LDS ordering; a workgroup barrier
// is needed only where the next step reads entries another wave wrote (15 of the 78 steps at 4096
// entries - a barrier-separated phase of 16 waves costs ~0.5 us here, the rest ~0.15 us).
template <bool DESC>
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long *keys, int p2) {
  for (int k = 2; k <= p2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (p2 >> 1); t += blockDim.x) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), o = i | j;
        const unsigned long long a = keys[i], b = keys[o];
        const bool first_half = (i & k) == 0;
        if ((first_half == DESC) ? a < b : a > b) {
          keys[i] = b;
          keys[o] = a;
        }
      }
      // next step: j / 2, or k (the first step of the next stage) after j == 1
      const int next_j = j > 1 ? (j >> 1) : k;
      if (j > 64 || next_j > 64 || (k == p2 && j == 1)) {
        __syncthreads();
      } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
}

__device__ __forceinline__ int next_pow2(int v) {
  int p = 128;
  while (p < v) p <<= 1;
  return p;
}

// The keys of a frame's candidates that count lie in [key_lo, key_hi): responses above thr = max * quality
// (lk_corner_select sends only those; lk_corner_response_nms also sends weaker 3x3 maxima) up to the maximum.
__device__ __forceinline__ void key_range(float eig_max, float quality, CornerKey &key_lo, CornerKey &key_hi) {
  const float top = fmaxf(eig_max, 0.f);
  const float thr = fmaxf(top * quality, 0.f);
  // STRICTLY above the threshold (THRESH_TOZERO keeps values > thr; responses are positive, so the next bit pattern is
  // the next value): lk_corner_select sends nothing else, the fused response pass sends every positive 3x3 maximum
  key_lo = (static_cast<CornerKey>(__float_as_uint(thr)) + 1ull) << 32;
  key_hi = (static_cast<CornerKey>(__float_as_uint(top)) + 1ull) << 32;
}
// The maximum response: from stats[] (eig_max), or - the fused response pass of the resident estimate, which has no
// finishing step - folded here from the 64 words of the frame's statistic slots (eig_slots: order-preserving keys,
// lk.hip slot_max); every wave does the fold itself.
__device__ __forceinline__ float load_eig_max(const float *eig_max, const unsigned *eig_slots) {
  if (eig_slots == nullptr) return *eig_max;
  unsigned v = eig_slots[threadIdx.x & 63];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned o = static_cast<unsigned>(__shfl_xor(static_cast<int>(v), d));
    v = o > v ? o : v;
  }
  return __uint_as_float((v >> 31) ? (v & 0x7fffffffu) : ~v);
}
__device__ __forceinline__ int bin_shift(CornerKey rlo, CornerKey rhi) {
  int sh = 0;
  while (((rhi - 1ull - rlo) >> sh) >= static_cast<unsigned long long>(kOrdBins)) ++sh;
  return sh;
}

// One wave: from the histogram of [rlo, rhi) (kOrdBins bins, LDS or global) and the number of
// candidates `above` it, the largest bin whose suffix count no longer fits a chunk.
//  over < 0: everything left fits (`fits` of them);  otherwise `fits` candidates lie above bin `over`.
__device__ __forceinline__ void chunk_cut(const int *hist, int above, int lane, int ccap, int &over_out, int &fits_out) {
  constexpr int kPer = kOrdBins / 64;
  int tot = 0;
  for (int q = 0; q < kPer; ++q) tot += hist[lane * kPer + q];
  int incl = tot;  // -> sum over lanes >= this one
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int dn = __shfl_down(incl, d);
    if (lane + d < 64) incl += dn;
  }
  int run = incl - tot + above;
  int over = -1, over_next = 0;  // bin, and the suffix count of the bins above it
  for (int q = kPer - 1; q >= 0; --q) {
    const int before = run;
    run += hist[lane * kPer + q];
    if (run > ccap && over < 0) {
      over = lane * kPer + q;
      over_next = before;
    }
  }
  int best = over;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d));
  const int all = __shfl(incl, 0) + above;
  // the owner of `best` broadcasts its suffix count
  const unsigned long long owner = __ballot(best >= 0 && over == best);
  const int src = owner ? __ffsll(static_cast<long long>(owner)) - 1 : 0;
  over_out = best;
  fits_out = best < 0 ? all : __shfl(over_next, src);
}

// ---------------------------------------------------------------------------------------------
// the head of the walking order, selected by many workgroups (the candidate list can hold
// millions of keys; a single workgroup streams it at the latency of one compute unit)
// ---------------------------------------------------------------------------------------------
constexpr int kHeadSegs = 4;  // consecutive chunks of the walking order prepared by the many-workgroup pass
struct OrderHeader {
  CornerKey floor_key[kHeadSegs];  // segment s of head[] holds every candidate key in [floor_key[s], floor_key[s-1])
  int count[kHeadSegs];            // their numbers
  int fill[kHeadSegs];             // reservation counters of corner_gather
  int nseg;                        // segments that are valid; the walk selects further chunks itself
  int walk[3];                     // written by corner_order: chunks taken, candidates in them, ordered batches
  int phase_us[6];                 // ... and where its time went: load, sort, coordinates, block tests, batches, total
  int sub_us[4];                   // inside 'batches': block hash build, first round, later rounds, appending
};

constexpr int kPreThreads = 256;

__global__ __launch_bounds__(kPreThreads) void corner_hist(const CornerKey *__restrict__ raw,
                                                           const int *__restrict__ raw_count, int cap,
                                                           const float *__restrict__ eig_max, float quality,
                                                           int *__restrict__ hist, const unsigned *__restrict__ eig_slots,
                                                           int count_bias) {
  __shared__ int s_hist[kOrdBins];
  for (int i = threadIdx.x; i < kOrdBins; i += kPreThreads) s_hist[i] = 0;
  __syncthreads();
  const int nkeys = min(max(*raw_count, 0) + count_bias, cap);
  CornerKey key_lo, key_hi;
  key_range(load_eig_max(eig_max, eig_slots), quality, key_lo, key_hi);
  const int sh = bin_shift(key_lo, key_hi);
  for (int i = blockIdx.x * kPreThreads + threadIdx.x; i < nkeys; i += gridDim.x * kPreThreads) {
    const CornerKey k = raw[i];
    if (k >= key_lo && k < key_hi) atomicAdd(&s_hist[static_cast<int>((k - key_lo) >> sh)], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kOrdBins; i += kPreThreads)
    if (s_hist[i]) atomicAdd(&hist[i], s_hist[i]);
}

__global__ __launch_bounds__(kPreThreads) void corner_gather(const CornerKey *__restrict__ raw,
                                                             const int *__restrict__ raw_count, int cap,
                                                             const float *__restrict__ eig_max, float quality,
                                                             const int *__restrict__ hist,
                                                             CornerKey *__restrict__ head,
                                                             OrderHeader *__restrict__ hdr, int ccap,
                                                             const unsigned *__restrict__ eig_slots, int count_bias) {
  __shared__ int s_suffix[kOrdBins + 1];  // candidates in bins >= c
  __shared__ int s_cut[kHeadSegs + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nkeys = min(max(*raw_count, 0) + count_bias, cap);
  CornerKey key_lo, key_hi;
  key_range(load_eig_max(eig_max, eig_slots), quality, key_lo, key_hi);
  const int sh = bin_shift(key_lo, key_hi);
  // ---- suffix counts of the histogram (every workgroup repeats this small computation) ----------
  if (wave == 0) {
    constexpr int kPer = kOrdBins / 64;
    int tot = 0;
    for (int q = 0; q < kPer; ++q) tot += hist[lane * kPer + q];
    int incl = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int dn = __shfl_down(incl, d);
      if (lane + d < 64) incl += dn;
    }
    int run = incl - tot;
    for (int q = kPer - 1; q >= 0; --q) {
      run += hist[lane * kPer + q];
      s_suffix[lane * kPer + q] = run;
    }
    if (lane == 0) {
      s_suffix[kOrdBins] = 0;
      s_cut[0] = kOrdBins;
    }
  }
  __syncthreads();
  // ---- segment s = bins [cut[s+1], cut[s]): as many whole bins as fit a chunk -------------------
  int nseg = 0;
  for (int sgm = 0; sgm < kHeadSegs; ++sgm) {
    const int prev = s_cut[sgm], base = s_suffix[prev];
    if (tid == 0) s_cut[sgm + 1] = prev;
    __syncthreads();
    // bins [c, prev) fit for every c from some bin on: the thread that sees the change stores it
    for (int c = tid; c < prev; c += kPreThreads)
      if (s_suffix[c] - base <= ccap && (c == 0 || s_suffix[c - 1] - base > ccap)) s_cut[sgm + 1] = c;
    __syncthreads();
    if (s_cut[sgm + 1] == prev) break;  // the next bin alone overfills a chunk: the walk refines it
    nseg = sgm + 1;
    if (s_cut[sgm + 1] == 0) break;  // the bottom of the key range
  }
  if (blockIdx.x == 0 && tid == 0) {
    hdr->nseg = nseg;
    for (int sgm = 0; sgm < nseg; ++sgm) {
      const int c = s_cut[sgm + 1];
      hdr->floor_key[sgm] = c == 0 ? key_lo : key_lo + (static_cast<CornerKey>(c) << sh);
      hdr->count[sgm] = s_suffix[c] - s_suffix[s_cut[sgm]];
    }
  }
  // ---- this workgroup's slice of the list: count per segment, reserve once per segment, write -----
  if (nseg == 0) return;
  CornerKey floor_key[kHeadSegs];
#pragma unroll
  for (int sgm = 0; sgm < kHeadSegs; ++sgm) {
    const int c = s_cut[min(sgm, nseg - 1) + 1];
    floor_key[sgm] = sgm < nseg ? (c == 0 ? key_lo : key_lo + (static_cast<CornerKey>(c) << sh)) : ~0ull;
  }
  auto segment_of = [&](CornerKey k) {  // kHeadSegs: not in the head
    if (k >= key_hi || k == 0ull) return kHeadSegs;
#pragma unroll
    for (int sgm = 0; sgm < kHeadSegs; ++sgm)
      if (k >= floor_key[sgm]) return sgm;
    return kHeadSegs;
  };
  const int per = (nkeys + gridDim.x - 1) / gridDim.x;
  const int s0 = min(blockIdx.x * per, nkeys), s1 = min(s0 + per, nkeys);
  int mine[kHeadSegs] = {0, 0, 0, 0};
  for (int i = s0 + tid; i < s1; i += kPreThreads) {
    const int sgm = segment_of(raw[i]);
#pragma unroll
    for (int q = 0; q < kHeadSegs; ++q) mine[q] += sgm == q ? 1 : 0;
  }
  __shared__ int s_wave4[kPreThreads / 64][kHeadSegs];
  __shared__ int s_base4[kHeadSegs];
  int before[kHeadSegs];
#pragma unroll
  for (int q = 0; q < kHeadSegs; ++q) {
    int incl = mine[q];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (lane >= d) incl += up;
    }
    if (lane == 63) s_wave4[wave][q] = incl;
    before[q] = incl - mine[q];
  }
  __syncthreads();
  if (tid < kHeadSegs) {
    int total = 0;
    for (int w = 0; w < kPreThreads / 64; ++w) total += s_wave4[w][tid];
    s_base4[tid] = total > 0 ? atomicAdd(&hdr->fill[tid], total) : 0;
  }
  __syncthreads();
  int at[kHeadSegs];
#pragma unroll
  for (int q = 0; q < kHeadSegs; ++q) {
    at[q] = s_base4[q] + before[q];
    for (int w = 0; w < wave; ++w) at[q] += s_wave4[w][q];
  }
  for (int i = s0 + tid; i < s1; i += kPreThreads) {
    const CornerKey k = raw[i];
    const int sgm = segment_of(k);
#pragma unroll
    for (int q = 0; q < kHeadSegs; ++q) {
      if (sgm == q) {
        if (at[q] < kChunkCap) head[static_cast<size_t>(q) * kChunkCap + at[q]] = k;
        ++at[q];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// goodFeaturesToTrack: ordered min-distance acceptance
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned cell_hash(unsigned cellkey) {
  return ((cellkey & 0xffffu) * 0x9E3779B1u ^ (cellkey >> 16) * 0x85EBCA6Bu) >> (32 - 13);
}
static_assert(kHashSlots == 1 << 13, "cell_hash delivers 13 bits");

// new head of a chain whose heads are 16-bit LDS entries; returns the previous head (LDS atomics are
// 32 bits wide: compare-and-swap on the word that holds the entry)
__device__ __forceinline__ short push_front(short *head, int value) {
  unsigned *word = reinterpret_cast<unsigned *>(reinterpret_cast<uintptr_t>(head) & ~static_cast<uintptr_t>(3));
  const int shift = (reinterpret_cast<uintptr_t>(head) & 2) ? 16 : 0;
  unsigned seen = *word;
  for (;;) {
    const unsigned want = (seen & ~(0xffffu << shift)) | ((static_cast<unsigned>(value) & 0xffffu) << shift);
    const unsigned prev = atomicCAS(word, seen, want);
    if (prev == seen) return static_cast<short>((seen >> shift) & 0xffffu);
    seen = prev;
  }
}

__global__ __launch_bounds__(kOrdThreads) void corner_order(const CornerKey *__restrict__ raw,
                                                            const int *__restrict__ raw_count, int cap,
                                                            const float *__restrict__ eig_max, float quality,
                                                            const CornerKey *__restrict__ head,
                                                            OrderHeader *__restrict__ hdr, int n, int cell,
                                                            unsigned md2_ceil, int use_grid, int max_corners, int ccap,
                                                            float2 *__restrict__ points, int *__restrict__ npoints,
                                                            const unsigned *__restrict__ eig_slots, int count_bias,
                                                            float *__restrict__ eig_max_out) {
  __shared__ CornerKey s_keys[kChunkCap];
  __shared__ int s_hist[kOrdBins];
  // the block's own candidates by cell (fixed point of the acceptance among the survivors)
  constexpr int kBlockSlots = kHashSlots;
  __shared__ short s_bhead[kBlockSlots];
  __shared__ short s_bnext[kOrdThreads];
  __shared__ uint2 s_bxy[kOrdThreads];  // x | y << 16, x cell | y cell << 16
  __shared__ unsigned char s_state[kOrdThreads];
  // second key buffer of the counting sort; afterwards the same memory holds, for the ordered chunk,
  // x | y << 16 (s_xy) and x cell | y cell << 16 (s_cl): the divisions are done once per chunk
  __shared__ CornerKey s_tmp[kChunkCap];
  unsigned *const s_xy = reinterpret_cast<unsigned *>(s_tmp);
  unsigned *const s_cl = s_xy + kChunkCap;
  __shared__ int s_off[kOrdBins];
  __shared__ int s_maxbin;
  __shared__ uint2 s_acc[kMaxCornersDev];   // accepted corners, same packing
  __shared__ short s_next[kMaxCornersDev];  // next accepted corner of the same cell (-1: none)
  // Accepted corners by cell: bucket = hash of the cell key, chained through s_next.  The buckets carry
  // no key: a chain may mix cells, its members are told apart by the cell key kept with each corner.
  // A lookup is then ONE read (the bucket's head) instead of a probe sequence whose length a wave
  // pays as the maximum over its 64 lanes, and the nine heads of a 3x3 neighbourhood are read together.
  __shared__ short s_hhead[kHashSlots];     // newest accepted corner of the bucket (-1: none)
  __shared__ unsigned short s_surv[kOrdThreads];  // survivors of the block test, in walking order
  __shared__ int s_wcount[kOrdWaves];
  __shared__ int s_fill, s_over, s_fits, s_nsurv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nkeys = min(max(*raw_count, 0) + count_bias, cap);
  CornerKey key_lo, upper;  // candidates not yet walked lie in [key_lo, upper)
  const float top_response = load_eig_max(eig_max, eig_slots);
  if (eig_max_out != nullptr && tid == 0) *eig_max_out = top_response;  // (the fused response pass leaves stats[] to this kernel)
  key_range(top_response, quality, key_lo, upper);
  for (int i = tid; i < kHashSlots; i += kOrdThreads) s_hhead[i] = -1;
  int nacc = 0;
  int remaining = nkeys;
  int st_chunks = 0, st_walked = 0, st_batches = 0;  // statistics of the walk (PYSTEPS_HIP_TRACE)
  long long tk_load = 0, tk_sort = 0, tk_xy = 0, tk_test = 0, tk_batch = 0;  // 100 MHz ticks per phase
  long long tk_sub[4] = {0, 0, 0, 0};
  long long tk_sub_mark = 0;
  auto sub_lap = [&](int which) {
    const long long now = wall_clock64();
    tk_sub[which] += now - tk_sub_mark;
    tk_sub_mark = now;
  };
  const long long tk_start = wall_clock64();
  long long tk_mark = tk_start;
  auto lap = [&](long long &acc) {
    const long long now = wall_clock64();
    acc += now - tk_mark;
    tk_mark = now;
  };
  const int head_segs = hdr != nullptr ? min(max(hdr->nseg, 0), kHeadSegs) : 0;
  int seg = 0;
  __syncthreads();
  while (remaining > 0 && nacc < max_corners) {
    CornerKey T = key_lo;
    int cnt = 0;
    if (seg < head_segs) {
      // ---- the first chunks: selected and gathered by corner_hist / corner_gather ----------------
      T = hdr->floor_key[seg];
      cnt = min(max(hdr->count[seg], 0), kChunkCap);
      const CornerKey *src = head + static_cast<size_t>(seg) * kChunkCap;
      for (int i = tid; i < cnt; i += kOrdThreads) s_keys[i] = src[i];
      ++seg;
    } else {
      // ---- threshold key T: the chunk is every candidate in [T, upper) ------------------------
      CornerKey rlo = key_lo, rhi = upper;
      int above = 0;  // candidates in [rhi, upper): part of the chunk whatever happens below
      for (int level = 0;; ++level) {  // (the bin width shrinks by 2^10 per level: at most 7 levels)
        const int sh = bin_shift(rlo, rhi);
        for (int i = tid; i < kOrdBins; i += kOrdThreads) s_hist[i] = 0;
        __syncthreads();
        for (int i0 = 0; i0 < nkeys; i0 += 8 * kOrdThreads) {  // eight loads in flight per thread
          CornerKey k[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * kOrdThreads + tid;
            k[u] = i < nkeys ? raw[i] : 0ull;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k[u] >= rlo && k[u] < rhi && k[u] != 0ull) atomicAdd(&s_hist[static_cast<int>((k[u] - rlo) >> sh)], 1);
        }
        __syncthreads();
        if (wave == 0) {
          int over, fits;
          chunk_cut(s_hist, above, lane, ccap, over, fits);
          if (lane == 0) {
            s_over = over;
            s_fits = fits;
          }
        }
        __syncthreads();
        const int over = s_over, fits = s_fits;  // `fits` candidates lie above bin `over`
        __syncthreads();
        if (over < 0 || level >= 8) {
          T = rlo;
          break;
        }
        if (fits >= ccap - (ccap >> 2) - (ccap >> 3)) {  // a chunk holds at least 5/8 of its cap (unless fewer are left)
          T = rlo + (static_cast<CornerKey>(over + 1) << sh);
          break;
        }
        // too few above the overfull bin: they are taken, the threshold is looked for inside the bin
        above = fits;
        rlo = rlo + (static_cast<CornerKey>(over) << sh);
        const CornerKey bin_hi = rlo + (1ull << sh);
        rhi = bin_hi < rhi ? bin_hi : rhi;
      }
      // ---- gather the chunk into LDS --------------------------------------------------------------
      if (tid == 0) s_fill = 0;
      __syncthreads();
      for (int i0 = 0; i0 < nkeys; i0 += 8 * kOrdThreads) {
        CornerKey k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + u * kOrdThreads + tid;
          k[u] = i < nkeys ? raw[i] : 0ull;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool keep = k[u] >= T && k[u] < upper && k[u] != 0ull;
          const unsigned long long mask = __ballot(keep);
          if (mask == 0ull) continue;
          int base = 0;
          if (lane == 0) base = atomicAdd(&s_fill, __popcll(mask));
          base = __shfl(base, 0);
          const int at = base + __popcll(mask & ((1ull << lane) - 1ull));
          if (keep && at < kChunkCap) s_keys[at] = k[u];
        }
      }
      __syncthreads();
      cnt = min(s_fill, kChunkCap);
    }
    if (cnt == 0) {  // an empty segment (whole bins only): on to the next chunk
      if (T <= key_lo) break;
      upper = T;
      __syncthreads();
      continue;
    }
    ++st_chunks;
    st_walked += cnt;
    lap(tk_load);
    // ---- order the chunk: counting sort over 1024 bins of [T, upper) (descending), the few keys
    // of a bin by one thread each; chunks with a crowded bin (many equal responses) take the
    // bitonic network instead ----------------------------------------------------------------------
    {
      const int sh2 = bin_shift(T, upper);
      for (int i = tid; i < kOrdBins; i += kOrdThreads) s_hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < cnt; i += kOrdThreads) atomicAdd(&s_hist[static_cast<int>((s_keys[i] - T) >> sh2)], 1);
      __syncthreads();
      if (wave == 0) {  // start of bin b in descending order = keys in the bins above it
        constexpr int kPer = kOrdBins / 64;
        int tot = 0, big = 0;
        for (int q = 0; q < kPer; ++q) {
          const int h = s_hist[lane * kPer + q];
          tot += h;
          big = max(big, h);
        }
        int incl = tot;  // -> sum over lanes >= this one
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int dn = __shfl_down(incl, d);
          if (lane + d < 64) incl += dn;
        }
        int run = incl - tot;
        for (int q = kPer - 1; q >= 0; --q) {
          s_off[lane * kPer + q] = run;
          run += s_hist[lane * kPer + q];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) big = max(big, __shfl_xor(big, d));
        if (lane == 0) s_maxbin = big;
      }
      __syncthreads();
      if (s_maxbin <= 1024) {
        // every thread keeps its (up to four) keys and their bins; the keys are scattered into their
        // bins' ranges of s_tmp in any order, then every key counts the larger keys of its bin - its
        // rank there - and goes to its final place in s_keys (a bin holds a handful of keys as a
        // rule; a crowded one costs its keys' owners a longer loop, not a different algorithm)
        constexpr int kMine = kChunkCap / kOrdThreads;
        CornerKey mine[kMine];
        int mine_bin[kMine];
#pragma unroll
        for (int q = 0; q < kMine; ++q) {
          const int i = tid + q * kOrdThreads;
          mine[q] = i < cnt ? s_keys[i] : 0ull;
          mine_bin[q] = i < cnt ? static_cast<int>((mine[q] - T) >> sh2) : -1;
        }
        for (int i = tid; i < kOrdBins; i += kOrdThreads) s_hist[i] = 0;  // now the fill cursors
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kMine; ++q)
          if (mine_bin[q] >= 0) s_tmp[s_off[mine_bin[q]] + atomicAdd(&s_hist[mine_bin[q]], 1)] = mine[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kMine; ++q) {
          if (mine_bin[q] < 0) continue;
          const int lo = s_off[mine_bin[q]], hi = lo + s_hist[mine_bin[q]];
          int rank = 0;
          for (int j = lo; j < hi; ++j) rank += s_tmp[j] > mine[q] ? 1 : 0;
          s_keys[lo + rank] = mine[q];
        }
        __syncthreads();
      } else {
        const int p2 = next_pow2(cnt);
        for (int i = cnt + tid; i < p2; i += kOrdThreads) s_keys[i] = 0ull;  // sorts behind every real key
        __syncthreads();
        bitonic_sort_lds<true>(s_keys, p2);
      }
    }
    lap(tk_sort);
    for (int i = tid; i < cnt; i += kOrdThreads) {
      const unsigned addr = static_cast<unsigned>(s_keys[i] & 0xffffffffull);
      const unsigned px = addr % static_cast<unsigned>(n), py = addr / static_cast<unsigned>(n);
      s_xy[i] = px | (py << 16);
      s_cl[i] = (px / static_cast<unsigned>(cell)) | ((py / static_cast<unsigned>(cell)) << 16);
    }
    __syncthreads();
    lap(tk_xy);
    // ---- walk the chunk: 1024 candidates at a time are first tested against the corners accepted
    // so far by all 16 waves at once (late in the walk that removes most of them), the survivors are
    // then decided in order in batches of 64 ----------------------------------------------------------
    for (int sb0 = 0; sb0 < cnt && nacc < max_corners; sb0 += kOrdThreads) {
      {
        const int i = sb0 + tid;
        const bool there = i < cnt;
        bool gone = false;
        if (use_grid && there && nacc > 0) {
          const unsigned xy = s_xy[i], cl = s_cl[i];
          const int x = static_cast<int>(xy & 0xffffu), y = static_cast<int>(xy >> 16);
          const int cx = static_cast<int>(cl & 0xffffu), cy = static_cast<int>(cl >> 16);
          unsigned want[9];
          int head[9];
#pragma unroll
          for (int nb = 0; nb < 9; ++nb) {  // the nine bucket heads: independent reads, one wait
            const int ncx = cx + (nb % 3) - 1, ncy = cy + (nb / 3) - 1;
            want[nb] = static_cast<unsigned>(ncx) | (static_cast<unsigned>(ncy) << 16);
            head[nb] = (ncx >= 0 && ncy >= 0) ? s_hhead[cell_hash(want[nb])] : -1;
          }
          // the nine chains side by side: one step of each per iteration (their reads are independent),
          // so a wave pays the LONGEST chain of its lanes once, not once per neighbour cell
          for (;;) {
            bool any = false;
#pragma unroll
            for (int nb = 0; nb < 9; ++nb) {
              const int q = head[nb];
              if (q < 0) continue;
              any = true;
              const uint2 a = s_acc[q];
              head[nb] = s_next[q];
              const unsigned dx = static_cast<unsigned>(abs(x - static_cast<int>(a.x & 0xffffu)));
              const unsigned dy = static_cast<unsigned>(abs(y - static_cast<int>(a.x >> 16)));
              if (a.y == want[nb] && dx * dx + dy * dy < md2_ceil) gone = true;  // neighbouring cells: < 2^31
            }
            if (!any || gone) break;
          }
        }
        const unsigned long long keep = __ballot(there && !gone);
        if (lane == 0) s_wcount[wave] = __popcll(keep);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < kOrdWaves; ++w) {
          if (w < wave) before += s_wcount[w];
          total += s_wcount[w];
        }
        if (there && !gone) s_surv[before + __popcll(keep & ((1ull << lane) - 1ull))] = static_cast<unsigned short>(i);
        if (tid == 0) s_nsurv = total;
        __syncthreads();
      }
      const int nsurv = s_nsurv;
      lap(tk_test);
      // ---- the survivors among themselves: the walk accepts a candidate unless an EARLIER accepted one
      // is too close, i.e. the accepted set is the greedy independent set of the conflict graph in
      // walking order.  Decided as a fixed point instead of one by one: a survivor with an accepted
      // earlier neighbour is rejected, one whose earlier neighbours are all rejected is accepted, the
      // others wait for the next round (states only move undecided -> decided, so reading a state
      // another thread is just writing is harmless; the first undecided survivor is decided in every
      // round, and at the reference's corner densities nearly all of them in the first).  Neighbours
      // are found through a hash of the block's own cells, like the accepted corners above.
      tk_sub_mark = wall_clock64();
      if (use_grid) {
        for (int q = tid; q < kBlockSlots; q += kOrdThreads) s_bhead[q] = -1;
        __syncthreads();
      }
      const bool mine = tid < nsurv;
      const int ci = mine ? s_surv[tid] : 0;
      const unsigned xy = mine ? s_xy[ci] : 0u, cl = mine ? s_cl[ci] : 0u;
      const int x = static_cast<int>(xy & 0xffffu), y = static_cast<int>(xy >> 16);
      const int cx = static_cast<int>(cl & 0xffffu), cy = static_cast<int>(cl >> 16);
      if (use_grid && mine) {
        s_bxy[tid] = make_uint2(xy, cl);
        s_bnext[tid] = push_front(&s_bhead[cell_hash(cl)], tid);
      }
      enum : unsigned char { kUndecided = 0, kAccepted = 1, kRejected = 2 };
      s_state[tid] = mine ? (use_grid ? kUndecided : kAccepted) : kRejected;
      __syncthreads();
      sub_lap(0);
      if (use_grid) {
        // No barrier between the rounds: a wave keeps re-evaluating its undecided candidates until none is left
        // (the states live in LDS, which has no cache - a volatile read sees what any wave of the workgroup has
        // written).  A candidate only waits for EARLIER candidates, the earliest undecided one never waits, and
        // the lanes of a wave advance together: every wave terminates, the fixed point is the one the rounds with
        // workgroup barriers reached (seven of them at ~2 us each on the bench frames).
        volatile unsigned char *state = s_state;
        bool undecided = mine;
        // the earlier neighbours that were still undecided at the full evaluation: all a later evaluation has to look at
        // (an accepted one was a rejection on the spot, a rejected one never matters again); more than two: evaluate in full
        int waits_for[2] = {-1, -1};
        bool memo = false;
        while (__ballot(undecided) != 0ull) {
          ++st_batches;  // (evaluations of this wave)
          if (undecided && memo) {
            bool rejected = false, blocked = false;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              if (waits_for[k] >= 0) {
                const unsigned char st = state[waits_for[k]];
                rejected = rejected || st == kAccepted;
                blocked = blocked || st == kUndecided;
              }
            }
            if (rejected) {
              state[tid] = kRejected;
              undecided = false;
            } else if (!blocked) {
              state[tid] = kAccepted;
              undecided = false;
            }
          } else if (undecided) {
            bool rejected = false, blocked = false;
            int n_wait = 0;
            unsigned want[9];
            int head[9];
#pragma unroll
            for (int nb = 0; nb < 9; ++nb) {
              const int ncx = cx + (nb % 3) - 1, ncy = cy + (nb / 3) - 1;
              want[nb] = static_cast<unsigned>(ncx) | (static_cast<unsigned>(ncy) << 16);
              head[nb] = (ncx >= 0 && ncy >= 0) ? s_bhead[cell_hash(want[nb])] : -1;
            }
            for (;;) {  // the nine chains side by side, as in the block test
              bool any = false;
#pragma unroll
              for (int nb = 0; nb < 9; ++nb) {
                const int q = head[nb];
                if (q < 0) continue;
                any = true;
                const uint2 a = s_bxy[q];
                const unsigned char st = state[q];
                head[nb] = s_bnext[q];
                const unsigned dx = static_cast<unsigned>(abs(x - static_cast<int>(a.x & 0xffffu)));
                const unsigned dy = static_cast<unsigned>(abs(y - static_cast<int>(a.x >> 16)));
                // (q >= tid: later in the walk, or the candidate itself)
                if (q < tid && a.y == want[nb] && dx * dx + dy * dy < md2_ceil) {  // neighbouring cells: < 2^31
                  rejected = rejected || st == kAccepted;
                  if (st == kUndecided) {
                    blocked = true;
                    if (n_wait < 2) waits_for[n_wait] = q;
                    ++n_wait;
                  }
                }
              }
              if (!any || rejected) break;
            }
            if (rejected) {
              state[tid] = kRejected;
              undecided = false;
            } else if (!blocked) {
              state[tid] = kAccepted;
              undecided = false;
            } else {
              memo = n_wait <= 2;
              if (!memo) waits_for[0] = waits_for[1] = -1;
            }
          }
          // (a wave that still waits gives the SIMD to the waves it waits for: the issue arbiter must never have to
          // choose a spinning wave over the one whose decision would release it)
          if (__ballot(undecided) != 0ull) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        sub_lap(2);  // (all evaluations)
      }
      // ---- the accepted survivors, in walking order, behind the corners so far (those beyond
      // max_corners are dropped: acceptance only depends on earlier candidates) -------------------
      {
        const bool acc_me = mine && s_state[tid] == kAccepted;
        const unsigned long long am = __ballot(acc_me);
        if (lane == 0) s_wcount[wave] = __popcll(am);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < kOrdWaves; ++w) {
          if (w < wave) before += s_wcount[w];
          total += s_wcount[w];
        }
        const int at = nacc + before + __popcll(am & ((1ull << lane) - 1ull));
        if (acc_me && at < max_corners) {
          s_acc[at] = make_uint2(xy, cl);
          points[at] = make_float2(static_cast<float>(x), static_cast<float>(y));
          s_next[at] = push_front(&s_hhead[cell_hash(cl)], at);  // (the order inside a chain does not matter)
        }
        nacc = min(nacc + total, max_corners);
        __syncthreads();
        sub_lap(3);  // (appending)
      }
      lap(tk_batch);
    }
    remaining -= cnt;
    if (T <= key_lo) break;  // the chunk reached the bottom of the key range
    upper = T;
    __syncthreads();
  }
  if (tid == 0) {
    *npoints = nacc;
    if (hdr) {
      hdr->walk[0] = st_chunks;
      hdr->walk[1] = st_walked;
      hdr->walk[2] = st_batches;
      const long long tk[6] = {tk_load, tk_sort, tk_xy, tk_test, tk_batch, wall_clock64() - tk_start};
      for (int q = 0; q < 6; ++q) hdr->phase_us[q] = static_cast<int>(tk[q] / 100);
      for (int q = 0; q < 4; ++q) hdr->sub_us[q] = static_cast<int>(tk_sub[q] / 100);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// after the outlier test: filter, decluster, interpolator preamble
// ---------------------------------------------------------------------------------------------
constexpr int kFinMax = 8192;  // pooled vectors (max_corners x frame pairs), as in dense_lk.hip

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmin(v, __shfl_xor(v, d));
  return v;
}
__device__ __forceinline__ double wave_max_f64(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d));
  return v;
}

__global__ __launch_bounds__(kOrdThreads) void vectors_finish(const double2 *__restrict__ pool_xy,
                                                              const double2 *__restrict__ pool_uv,
                                                              const unsigned char *__restrict__ flags,
                                                              const int *__restrict__ pool_count, int capacity,
                                                              double scale, int m, int n,
                                                              float2 *__restrict__ out_xy,
                                                              float2 *__restrict__ out_uv,
                                                              IdwDyn *__restrict__ dyn) {
  __shared__ unsigned long long s_keys[kFinMax];  // (x cell << 16 | y cell) << 32 | sample index
  __shared__ unsigned short s_seg[kFinMax];       // output slot of the cell an ordered entry belongs to
  __shared__ int s_scan[kOrdWaves];
  __shared__ double s_stat[6][kOrdWaves];
  __shared__ double s_first[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pooled = min(max(*pool_count, 0), min(capacity, kFinMax));
  const bool decl = scale > 1.0;
  constexpr unsigned long long kDropped = ~0ull;
  // ---- keys: dropped vectors sort behind the kept ones ----------------------------------------
  const int p2 = next_pow2(pooled);
  for (int i = tid; i < p2; i += kOrdThreads) {
    unsigned long long key = kDropped;
    if (i < pooled && !(pooled >= 2 && flags[i])) {  // fewer than two samples: nothing is an outlier
      unsigned cellkey = 0u;
      if (decl) {
        const double2 p = pool_xy[i];
        const double fx = floor(p.x / scale), fy = floor(p.y / scale);
        const unsigned cx = static_cast<unsigned>(fmin(fmax(fx, 0.0), 65535.0));
        const unsigned cy = static_cast<unsigned>(fmin(fmax(fy, 0.0), 65535.0));
        cellkey = (cx << 16) | cy;
      }
      // without declustering every vector is its own "cell", in input order
      key = decl ? (static_cast<unsigned long long>(cellkey) << 32) | static_cast<unsigned>(i)
                 : static_cast<unsigned long long>(i) << 32 | static_cast<unsigned>(i);
    }
    s_keys[i] = key;
  }
  __syncthreads();
  bitonic_sort_lds<false>(s_keys, p2);
  // ---- cells: entry t opens a cell if its cell key differs from the entry before ---------------
  // each thread owns p2 / 1024 consecutive entries (at least 1 when p2 < 1024: guarded)
  const int per = max(p2 / kOrdThreads, 1);
  const int t0 = tid * per;
  int heads = 0, kept_here = 0;
  for (int q = 0; q < per; ++q) {
    const int t = t0 + q;
    if (t >= p2) break;
    const unsigned long long k = s_keys[t];
    if (k == kDropped) continue;
    ++kept_here;
    if (t == 0 || (s_keys[t - 1] >> 32) != (k >> 32)) ++heads;
  }
  // exclusive scan of `heads` over the threads
  int incl = heads;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  int before = incl - heads, cells = 0;
  for (int w = 0; w < kOrdWaves; ++w) {
    if (w < wave) before += s_scan[w];
    cells += s_scan[w];
  }
  {
    int slot = before - 1;
    for (int q = 0; q < per; ++q) {
      const int t = t0 + q;
      if (t >= p2) break;
      const unsigned long long k = s_keys[t];
      if (k == kDropped) continue;
      if (t == 0 || (s_keys[t - 1] >> 32) != (k >> 32)) ++slot;
      s_seg[t] = static_cast<unsigned short>(slot);
    }
  }
  (void)kept_here;
  __syncthreads();
  // ---- medians: every entry ranks itself within its cell; the lower median's owner writes -------
  double vmin = INFINITY, vmax = -INFINITY;
  double xmin = 0.0, xmax = static_cast<double>(n) - 1.0, ymin = 0.0, ymax = static_cast<double>(m) - 1.0;
  for (int t = tid; t < p2; t += kOrdThreads) {
    const unsigned long long k = s_keys[t];
    if (k == kDropped) continue;
    const unsigned cellkey = static_cast<unsigned>(k >> 32);
    int s = t, e = t + 1;
    while (s > 0 && static_cast<unsigned>(s_keys[s - 1] >> 32) == cellkey) --s;
    while (e < p2 && static_cast<unsigned>(s_keys[e] >> 32) == cellkey) ++e;
    const int members = e - s;
    const int me = static_cast<int>(k & 0xffffffffull);
    const double2 mxy = pool_xy[me], muv = pool_uv[me];
    const double mine[4] = {mxy.x, mxy.y, muv.x, muv.y};
    int rank[4] = {0, 0, 0, 0};
    double succ[4] = {INFINITY, INFINITY, INFINITY, INFINITY};  // smallest value ordered behind mine
    for (int q = s; q < e; ++q) {
      if (q == t) continue;
      const int o = static_cast<int>(s_keys[q] & 0xffffffffull);
      const double2 oxy = pool_xy[o], ouv = pool_uv[o];
      const double other[4] = {oxy.x, oxy.y, ouv.x, ouv.y};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool below = other[c] < mine[c] || (other[c] == mine[c] && q < t);
        rank[c] += below ? 1 : 0;
        if (!below) succ[c] = fmin(succ[c], other[c]);
      }
    }
    const int lower = (members - 1) / 2;
    const bool even = (members & 1) == 0;
    const int slot = s_seg[t];
    double med[4];
    bool have[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      have[c] = rank[c] == lower;
      med[c] = 0.5 * (mine[c] + (even ? succ[c] : mine[c]));  // np.median: mean of the two middle values
    }
    if (have[0]) {
      reinterpret_cast<float *>(out_xy)[2 * slot] = static_cast<float>(med[0]);
      xmin = fmin(xmin, med[0]);
      xmax = fmax(xmax, med[0]);
    }
    if (have[1]) {
      reinterpret_cast<float *>(out_xy)[2 * slot + 1] = static_cast<float>(med[1]);
      ymin = fmin(ymin, med[1]);
      ymax = fmax(ymax, med[1]);
    }
    if (have[2]) {
      reinterpret_cast<float *>(out_uv)[2 * slot] = static_cast<float>(med[2]);
      vmin = fmin(vmin, med[2]);
      vmax = fmax(vmax, med[2]);
      if (slot == 0) s_first[0] = med[2];
    }
    if (have[3]) {
      reinterpret_cast<float *>(out_uv)[2 * slot + 1] = static_cast<float>(med[3]);
      vmin = fmin(vmin, med[3]);
      vmax = fmax(vmax, med[3]);
      if (slot == 0) s_first[1] = med[3];
    }
  }
  // ---- statistics for the interpolator preamble -------------------------------------------------
  const double part[6] = {wave_min_f64(vmin), wave_max_f64(vmax), wave_min_f64(xmin),
                          wave_max_f64(xmax), wave_min_f64(ymin), wave_max_f64(ymax)};
  if (lane == 0)
    for (int c = 0; c < 6; ++c) s_stat[c][wave] = part[c];
  __syncthreads();
  if (tid == 0) {
    double r[6];
    for (int c = 0; c < 6; ++c) {
      r[c] = s_stat[c][0];
      for (int w = 1; w < kOrdWaves; ++w) r[c] = (c & 1) ? fmax(r[c], s_stat[c][w]) : fmin(r[c], s_stat[c][w]);
    }
    IdwDyn d;
    d.L = cells;
    d.mode = 0;
    d.cu = 0.f;
    d.cv = 0.f;
    d.reach = 1.f;
    if (cells == 0) {  // lucaskanade.py:245-249, 268-269: zero field
      d.mode = 1;
    } else if (cells == 1) {  // decorators.py:200-203: one sample -> constant field
      d.mode = 1;
      d.cu = static_cast<float>(s_first[0]);
      d.cv = static_cast<float>(s_first[1]);
    } else if (r[0] == r[1]) {  // decorators.py:207-208: all elements equal
      d.mode = 1;
      d.cu = static_cast<float>(s_first[0]);
      d.cv = static_cast<float>(s_first[0]);
    } else {
      const double dx = r[3] - r[2], dy = r[5] - r[4];
      d.reach = static_cast<float>(fma(sqrt(fma(dx, dx, dy * dy)), 1.001, 1.0));
    }
    *dyn = d;
  }
}

}  // namespace

int corner_order_max_corners() { return kMaxCornersDev; }

// workspace of one ordering: [histogram | header | head keys]
constexpr size_t kOrdOffHdr = kOrdBins * sizeof(int);
constexpr size_t kOrdOffHead = kOrdOffHdr + 128;
static_assert(sizeof(OrderHeader) <= 128, "header slot");
static_assert(offsetof(OrderHeader, phase_us) == offsetof(OrderHeader, walk) + 3 * sizeof(int), "walk statistics are one int[13]");
size_t corner_order_ws_bytes() { return kOrdOffHead + static_cast<size_t>(kHeadSegs) * kChunkCap * sizeof(CornerKey); }

size_t corner_order_clear_bytes() { return kOrdOffHead; }
size_t corner_order_walk_stats_offset() { return kOrdOffHdr + offsetof(OrderHeader, walk); }

bool corner_order_supported(int m, int n, double min_distance, int max_corners) {
  // LDS list of accepted corners, 16-bit coordinates, squared distances of neighbouring cells in 32 bits
  return max_corners <= kMaxCornersDev && m <= 65535 && n <= 65535 && min_distance < 16383.0;
}

hipError_t launch_corner_order(const unsigned long long *raw_dev, const int *raw_count_dev, int cap,
                               const float *eig_max_dev, float quality, int n, double min_distance,
                               int max_corners, void *ws_dev, float *points_dev, int *npoints_dev,
                               hipStream_t stream, int (*before_walk)(void *), void *before_walk_arg,
                               bool ws_is_cleared, const unsigned *eig_slots_dev, int count_bias, float *eig_max_out_dev) {
  const int cell = static_cast<int>(std::lrint(min_distance)) > 1 ? static_cast<int>(std::lrint(min_distance)) : 1;
  const double md2 = std::ceil(min_distance * min_distance);
  const unsigned md2_ceil = md2 >= 2147483647.0 ? 2147483647u : static_cast<unsigned>(md2);
  char *ws = static_cast<char *>(ws_dev);
  int *hist = reinterpret_cast<int *>(ws);
  OrderHeader *hdr = reinterpret_cast<OrderHeader *>(ws + kOrdOffHdr);
  CornerKey *head = reinterpret_cast<CornerKey *>(ws + kOrdOffHead);
  if (!ws_is_cleared) {  // (the corner entry points clear it from lk_max_final)
    const hipError_t e = hipMemsetAsync(ws, 0, kOrdOffHead, stream);
    if (e != hipSuccess) return e;
  }
  // the list is streamed by up to 128 workgroups (its length is only known on the device)
  const int groups = std::max(1, std::min(128, (cap + 4 * kPreThreads - 1) / (4 * kPreThreads)));
  hipLaunchKernelGGL(corner_hist, dim3(groups), dim3(kPreThreads), 0, stream, raw_dev, raw_count_dev, cap, eig_max_dev,
                     quality, hist, eig_slots_dev, count_bias);
  // candidates ordered at a time: the walk rarely needs more than ~2 x max_corners of them, and a chunk
  // that covers a narrower key range is ordered faster (finer bins of the counting sort)
  int ccap = 1024;
  while (ccap < 2 * max_corners && ccap < kChunkCap) ccap <<= 1;
  hipLaunchKernelGGL(corner_gather, dim3(groups), dim3(kPreThreads), 0, stream, raw_dev, raw_count_dev, cap, eig_max_dev,
                     quality, hist, head, hdr, ccap, eig_slots_dev, count_bias);
  // the walk is a single workgroup: whatever the caller can run beside it is forked off here
  if (before_walk != nullptr && before_walk(before_walk_arg) != 0) return hipErrorUnknown;
  hipLaunchKernelGGL(corner_order, dim3(1), dim3(kOrdThreads), 0, stream, raw_dev, raw_count_dev, cap, eig_max_dev,
                     quality, head, hdr, n, cell, md2_ceil, min_distance >= 1.0 ? 1 : 0, max_corners, ccap,
                     reinterpret_cast<float2 *>(points_dev), npoints_dev, eig_slots_dev, count_bias, eig_max_out_dev);
  return hipGetLastError();
}

hipError_t launch_vectors_finish(const double *pool_xy_dev, const double *pool_uv_dev,
                                 const unsigned char *flags_dev, const int *pool_count_dev, int capacity,
                                 double decl_scale, int m, int n, float *xy_out_dev, float *uv_out_dev,
                                 IdwDyn *dyn_dev, hipStream_t stream) {
  hipLaunchKernelGGL(vectors_finish, dim3(1), dim3(kOrdThreads), 0, stream,
                     reinterpret_cast<const double2 *>(pool_xy_dev), reinterpret_cast<const double2 *>(pool_uv_dev),
                     flags_dev, pool_count_dev, capacity, decl_scale, m, n, reinterpret_cast<float2 *>(xy_out_dev),
                     reinterpret_cast<float2 *>(uv_out_dev), dyn_dev);
  return hipGetLastError();
}

}  // namespace psh

// ---------------------------------------------------------------------------------------------
// C ABI: the two kernels with host buffers (the staged Python loop, the row-band path that
// gathers candidate keys from several ranks, and the parity tests use these)
// ---------------------------------------------------------------------------------------------
extern "C" int psh_lk_order_host(const unsigned long long *keys_host, int count, float response_max,
                                 double quality_level, int m, int n, double min_distance, int max_corners,
                                 float *points_host, int *count_host) {
  PSH_REQUIRE_INIT();
  if (!points_host || !count_host || (count > 0 && !keys_host)) return psh::fail(PSH_EINVAL, "lk_order: NULL pointer");
  if (count < 0 || m <= 0 || n <= 0 || max_corners <= 0) return psh::fail(PSH_EINVAL, "lk_order: invalid argument");
  if (!psh::corner_order_supported(m, n, min_distance, max_corners))
    return psh::fail(PSH_EUNSUPPORTED, "lk_order: more than %d corners, 65535 rows / columns or min_distance >= 16383",
                     psh::kMaxCornersDev);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t key_bytes = (static_cast<size_t>(count) * sizeof(unsigned long long) + 255) & ~static_cast<size_t>(255);
  const size_t pts_bytes = static_cast<size_t>(max_corners) * 2 * sizeof(float);
  void *blk = nullptr;
  const size_t ws_bytes = (psh::corner_order_ws_bytes() + 255) & ~static_cast<size_t>(255);
  if (int rc = psh_malloc(&blk, key_bytes + 256 + pts_bytes + 256 + ws_bytes)) return rc;
  char *base = static_cast<char *>(blk);
  unsigned long long *d_keys = reinterpret_cast<unsigned long long *>(base);
  int *d_hdr = reinterpret_cast<int *>(base + key_bytes);  // [count | accepted | response max (float)]
  float *d_pts = reinterpret_cast<float *>(base + key_bytes + 256);
  void *d_ws = base + ((key_bytes + 256 + pts_bytes + 255) & ~static_cast<size_t>(255));
  int rc = PSH_OK;
  auto run = [&]() -> int {
    struct {
      int count, accepted;
      float top;
    } hdr = {count, 0, response_max};
    if (count > 0) PSH_HIP(hipMemcpyAsync(d_keys, keys_host, static_cast<size_t>(count) * 8, hipMemcpyHostToDevice, c.stream));
    PSH_HIP(hipMemcpyAsync(d_hdr, &hdr, sizeof(hdr), hipMemcpyHostToDevice, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));  // hdr lives on this stack frame
    PSH_HIP(psh::launch_corner_order(d_keys, d_hdr, count, reinterpret_cast<const float *>(d_hdr + 2),
                                     static_cast<float>(quality_level), n, min_distance, max_corners, d_ws, d_pts,
                                     d_hdr + 1, c.stream, nullptr, nullptr, false));
    int accepted = 0;
    PSH_HIP(hipMemcpyAsync(&accepted, d_hdr + 1, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    accepted = accepted < 0 ? 0 : accepted > max_corners ? max_corners : accepted;
    if (accepted > 0) {
      PSH_HIP(hipMemcpyAsync(points_host, d_pts, static_cast<size_t>(accepted) * 2 * sizeof(float), hipMemcpyDeviceToHost,
                             c.stream));
      PSH_HIP(hipStreamSynchronize(c.stream));
    }
    *count_host = accepted;
    return PSH_OK;
  };
  rc = run();
  (void)psh_free(blk);
  return rc;
}

extern "C" int psh_vectors_finish_host(const double *xy, const double *values, const unsigned char *outlier_flags,
                                       int count, double decl_scale, int m, int n, float *out_xy, float *out_values,
                                       int *out_count, int *out_mode, float *out_const, float *out_reach) {
  PSH_REQUIRE_INIT();
  if (count < 0 || m <= 0 || n <= 0) return psh::fail(PSH_EINVAL, "vectors_finish: invalid argument");
  if (count > psh::kFinMax) return psh::fail(PSH_EUNSUPPORTED, "vectors_finish: more than %d vectors", psh::kFinMax);
  if (!out_xy || !out_values || !out_count || !out_mode || !out_const || !out_reach ||
      (count > 0 && (!xy || !values || !outlier_flags)))
    return psh::fail(PSH_EINVAL, "vectors_finish: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t cap = static_cast<size_t>(count > 0 ? count : 1);
  const size_t vec = cap * 16, off_uv = vec, off_fl = 2 * vec, off_cnt = (off_fl + cap + 255) & ~static_cast<size_t>(255);
  const size_t off_oxy = off_cnt + 256, off_ouv = off_oxy + cap * 8, off_dyn = off_ouv + cap * 8;
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, off_dyn + sizeof(psh::IdwDyn))) return rc;
  char *base = static_cast<char *>(blk);
  auto run = [&]() -> int {
    if (count > 0) {
      PSH_HIP(hipMemcpyAsync(base, xy, vec, hipMemcpyHostToDevice, c.stream));
      PSH_HIP(hipMemcpyAsync(base + off_uv, values, vec, hipMemcpyHostToDevice, c.stream));
      PSH_HIP(hipMemcpyAsync(base + off_fl, outlier_flags, cap, hipMemcpyHostToDevice, c.stream));
    }
    PSH_HIP(hipMemcpyAsync(base + off_cnt, &count, sizeof(int), hipMemcpyHostToDevice, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    PSH_HIP(psh::launch_vectors_finish(reinterpret_cast<const double *>(base), reinterpret_cast<const double *>(base + off_uv),
                                       reinterpret_cast<const unsigned char *>(base + off_fl),
                                       reinterpret_cast<const int *>(base + off_cnt), static_cast<int>(cap), decl_scale, m, n,
                                       reinterpret_cast<float *>(base + off_oxy), reinterpret_cast<float *>(base + off_ouv),
                                       reinterpret_cast<psh::IdwDyn *>(base + off_dyn), c.stream));
    psh::IdwDyn d;
    PSH_HIP(hipMemcpyAsync(&d, base + off_dyn, sizeof(d), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    const int L = d.L < 0 ? 0 : d.L > count ? count : d.L;
    if (L > 0) {
      PSH_HIP(hipMemcpyAsync(out_xy, base + off_oxy, static_cast<size_t>(L) * 8, hipMemcpyDeviceToHost, c.stream));
      PSH_HIP(hipMemcpyAsync(out_values, base + off_ouv, static_cast<size_t>(L) * 8, hipMemcpyDeviceToHost, c.stream));
      PSH_HIP(hipStreamSynchronize(c.stream));
    }
    *out_count = L;
    *out_mode = d.mode;
    out_const[0] = d.cu;
    out_const[1] = d.cv;
    *out_reach = d.reach;
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  return rc;
}
