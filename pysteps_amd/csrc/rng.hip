// numpy.random.RandomState.randn on the device: MT19937 + the polar method of NumPy's legacy
// generator, for B independent streams (one per ensemble member).
//
// Why: the STEPS member loop draws one m x n field of white noise per member and time step from the
// member's RandomState (pysteps/noise/fftgenerators.py:400 `randstate.randn(...)`, generators seeded
// by the chain of pysteps/nowcasts/steps.py:885-898).  The stream is part of the reference's result,
// so a resident member loop has to reproduce it - on the host it costs 0.3 s per 4096^2 field and a
// 134 MB transfer.
//
// What NumPy does (third party, numpy 2.2: numpy/random/src/mt19937/mt19937.c,
// numpy/random/src/legacy/legacy-distributions.c `legacy_gauss`), restated:
//   next32: MT19937 word (state 624 words, twist x[k+624] = x[k+397] ^ A(upper(x[k]) | lower(x[k+1]))), tempered
//   double: a = next32 >> 5, b = next32 >> 6;  d = (a 2^26 + b) / 2^53
//   gauss : if a value is cached, return it; else repeat x1 = 2 d - 1, x2 = 2 d' - 1, r2 = x1 x1 + x2 x2
//           until 0 < r2 < 1;  f = sqrt(-2 log(r2) / r2);  cache f x1, return f x2
// An attempt always consumes four words, so attempt k of a draw sits at a fixed place of the word
// sequence and the draw is a stream compaction: output pair p comes from the p-th accepted attempt.
//
//   mt_produce   one workgroup extends a stream's ring of raw MT19937 words: the whole 624-word twist
//                in ONE barrier phase (every new word written as a function of the old block only: 3,
//                5 or 7 old words), 640 threads.  A stream is sequential - 0.4 us per twist, 27 ms for
//                the words of one 4096^2 field - so it is cut into chunks of 512 blocks whose start
//                states come from the generator's GF(2)-linear structure:
//   mt_jump      state J words ahead = g_J(A) state, g_J = x^J mod (characteristic polynomial of the
//                word transition A), evaluated by Horner 32 coefficients at a time (mt_jump_tables.h,
//                tools/gen_mt_jump.py); the start states of 2^l chunks in l rounds of doubling.  Every
//                chunk is then produced by its own workgroup (mt_produce_chunks); when the start states
//                are used up the grid is anchored anew at the last block produced (mt_anchor)
//   polar_count  accepted attempts per tile of 1024 attempts (a window of W attempts per draw; W holds
//                the pairs needed with > 10 sigma to spare)
//   polar_scan   one workgroup per stream: exclusive scan of the tile counts
//   polar_write  f per accepted attempt, out[2p] = f x2, out[2p+1] = f x1; the attempt that completes
//                the draw publishes the stream's new position / cached value
// Integer work (words, positions, accept / reject, the final state) is bit-identical with NumPy; the
// values are too wherever the C library's log() is correctly rounded (glibc: 99.9 % of the arguments),
// 1 ulp otherwise - cr_log.h.  Everything is asynchronous; the handle owns its stream, so a draw for
// the next time step can run beside the rest of the member loop (psh_rng_randn_dev side != 0).
#include <algorithm>
#include <vector>

#include "common.h"
#include "cr_log.h"
#include "mt_jump_tables.h"

namespace psh {
namespace {

constexpr int kMtN = 624;
constexpr int kProduceThreads = 640;
constexpr int kTileThreads = 256;
constexpr int kTileAttempts = 1024;  // 4 per thread
constexpr double kAccept = 0.78539816339744830962;  // pi / 4

__constant__ double c_log_table[PSH_CRLOG_N][3] = {PSH_CRLOG_TABLE};

struct RngDyn {
  unsigned long long pos;  // absolute index of the next unread word (word 0 = first word of the initial key block)
  double gauss;            // cached value (legacy_gauss), 0.0 if none
  int has_gauss;
  int err;  // 1: the window did not hold enough accepted attempts
};

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((b & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// ring: (streams, ring_blocks * 624) raw state words, block k of a stream in slot k % ring_blocks.
// Generates blocks first_block .. first_block + nblocks - 1 of every stream from block first_block - 1.
__global__ __launch_bounds__(kProduceThreads) void mt_produce(uint32_t *__restrict__ rings, unsigned ring_blocks,
                                                              unsigned long long first_block, int nblocks) {
  __shared__ uint32_t s_blk[2][kMtN];
  uint32_t *ring = rings + static_cast<size_t>(blockIdx.x) * ring_blocks * kMtN;
  const int i = threadIdx.x;
  {
    const uint32_t *src = ring + static_cast<size_t>((first_block - 1) % ring_blocks) * kMtN;
    if (i < kMtN) s_blk[0][i] = src[i];
  }
  __syncthreads();
  for (int k = 0; k < nblocks; ++k) {
    const uint32_t *o = s_blk[k & 1];
    uint32_t *nw = s_blk[(k & 1) ^ 1];
    if (i < kMtN) {
      uint32_t v;
      if (i < 227) {
        v = o[i + 397] ^ mt_mix(o[i], o[i + 1]);
      } else if (i < 454) {
        v = o[i + 170] ^ mt_mix(o[i - 227], o[i - 226]) ^ mt_mix(o[i], o[i + 1]);
      } else if (i < 623) {
        v = o[i - 57] ^ mt_mix(o[i - 454], o[i - 453]) ^ mt_mix(o[i - 227], o[i - 226]) ^ mt_mix(o[i], o[i + 1]);
      } else {
        const uint32_t n0 = o[397] ^ mt_mix(o[0], o[1]);
        const uint32_t n396 = o[566] ^ mt_mix(o[169], o[170]) ^ mt_mix(o[396], o[397]);
        v = n396 ^ mt_mix(o[623], n0);
      }
      nw[i] = v;
      __builtin_nontemporal_store(v, ring + static_cast<size_t>((first_block + k) % ring_blocks) * kMtN + i);
    }
    __syncthreads();
  }
}

constexpr int kChunkBlocks = PSH_MT_CHUNK_BLOCKS;
__device__ const uint32_t d_jump_table[PSH_MT_JUMP_LEVELS][PSH_MT_JUMP_WORDS] = {PSH_MT_JUMP_TABLE};

// bases: (streams, n_chunks, 624): bases[b][j] = the stream's window at word 624 * kChunkBlocks * j (block
// kChunkBlocks * j), exact except for the low 31 bits of its first word, which the recurrence never reads.
// One workgroup: bases[b][first_dst + x] = g_level(A) bases[b][first_dst + x - 2^level].
// Horner, 32 coefficients per round (chunk c of the polynomial = coefficients 32 c .. 32 c + 31):
//   acc <- A^32 acc  ^  XOR_b chunk_b A^b s0,      (A^b s0)[t] = E[b + t], E = s0 extended by 31 words
// acc lives in LDS as a ring with head h: advancing by 32 overwrites the 32 dropped words with the 32 new
// ones (all functions of old words, 32 <= 227) and leaves everything else in place.
__global__ __launch_bounds__(kProduceThreads) void mt_jump(uint32_t *__restrict__ bases, unsigned n_chunks, int level,
                                                           unsigned first_dst, unsigned count) {
  __shared__ uint32_t s_acc[kMtN];
  __shared__ uint32_t s_ext[kMtN + 32];
  if (blockIdx.x >= count) return;
  const unsigned dst_chunk = first_dst + blockIdx.x;
  uint32_t *stream_bases = bases + static_cast<size_t>(blockIdx.y) * n_chunks * kMtN;
  const uint32_t *src = stream_bases + static_cast<size_t>(dst_chunk - (1u << level)) * kMtN;
  uint32_t *dst = stream_bases + static_cast<size_t>(dst_chunk) * kMtN;
  const int t = threadIdx.x;
  if (t < kMtN) {
    s_ext[t] = src[t];
    s_acc[t] = 0u;
  }
  __syncthreads();
  if (t < 31) s_ext[kMtN + t] = s_ext[t + 397] ^ mt_mix(s_ext[t], s_ext[t + 1]);
  __syncthreads();
  const uint32_t *g = d_jump_table[level];
  int h = 0;
  auto slot = [](int i) { return i >= kMtN ? i - kMtN : i; };
  for (int c = kMtN - 1; c >= 0; --c) {
    uint32_t chunk = g[c];  // the same word in every lane
    uint32_t fresh = 0u, x = 0u;
    if (t >= kMtN - 32 && t < kMtN) {  // the word that enters the window at logical position t
      const int j = t - (kMtN - 32);
      fresh = s_acc[slot(slot(h + j) + 397)] ^ mt_mix(s_acc[slot(h + j)], s_acc[slot(slot(h + j) + 1)]);
    }
    if (t < kMtN) {
      while (chunk) {
        const int b = __builtin_ctz(chunk);
        x ^= s_ext[t + b];
        chunk &= chunk - 1u;
      }
    }
    __syncthreads();  // every read of the old window is done
    h = slot(h + 32);
    if (t < kMtN) {
      const int p = slot(h + t);
      s_acc[p] = (t < kMtN - 32 ? s_acc[p] : fresh) ^ x;
    }
    __syncthreads();
  }
  if (t < kMtN) dst[t] = s_acc[slot(h + t)];
}

// bases[b][0] = block `block` of every stream's ring (the state the chunk grid is anchored at)
__global__ __launch_bounds__(kProduceThreads) void mt_anchor(const uint32_t *__restrict__ rings, unsigned ring_blocks,
                                                             unsigned long long block, uint32_t *__restrict__ bases,
                                                             unsigned n_chunks) {
  const uint32_t *src = rings + (static_cast<size_t>(blockIdx.x) * ring_blocks + block % ring_blocks) * kMtN;
  uint32_t *dst = bases + static_cast<size_t>(blockIdx.x) * n_chunks * kMtN;
  if (threadIdx.x < kMtN) dst[threadIdx.x] = src[threadIdx.x];
}

// one workgroup per (chunk, stream): blocks anchor + chunk * C + 1 .. anchor + chunk * C + C from the chunk's start state
__global__ __launch_bounds__(kProduceThreads) void mt_produce_chunks(uint32_t *__restrict__ rings, unsigned ring_blocks,
                                                                     const uint32_t *__restrict__ bases, unsigned n_chunks,
                                                                     unsigned first_chunk, unsigned long long anchor_block) {
  __shared__ uint32_t s_blk[2][kMtN];
  const unsigned chunk = first_chunk + blockIdx.x;
  uint32_t *ring = rings + static_cast<size_t>(blockIdx.y) * ring_blocks * kMtN;
  const uint32_t *base = bases + (static_cast<size_t>(blockIdx.y) * n_chunks + chunk) * kMtN;
  const int i = threadIdx.x;
  if (i < kMtN) s_blk[0][i] = base[i];
  __syncthreads();
  const unsigned long long first_block = anchor_block + static_cast<unsigned long long>(chunk) * kChunkBlocks + 1;
  for (int k = 0; k < kChunkBlocks; ++k) {
    const uint32_t *o = s_blk[k & 1];
    uint32_t *nw = s_blk[(k & 1) ^ 1];
    if (i < kMtN) {
      uint32_t v;
      if (i < 227) {
        v = o[i + 397] ^ mt_mix(o[i], o[i + 1]);
      } else if (i < 454) {
        v = o[i + 170] ^ mt_mix(o[i - 227], o[i - 226]) ^ mt_mix(o[i], o[i + 1]);
      } else if (i < 623) {
        v = o[i - 57] ^ mt_mix(o[i - 454], o[i - 453]) ^ mt_mix(o[i - 227], o[i - 226]) ^ mt_mix(o[i], o[i + 1]);
      } else {
        const uint32_t n0 = o[397] ^ mt_mix(o[0], o[1]);
        const uint32_t n396 = o[566] ^ mt_mix(o[169], o[170]) ^ mt_mix(o[396], o[397]);
        v = n396 ^ mt_mix(o[623], n0);
      }
      nw[i] = v;
      __builtin_nontemporal_store(v, ring + static_cast<size_t>((first_block + k) % ring_blocks) * kMtN + i);
    }
    __syncthreads();
  }
}

struct Attempt {
  double x1, x2, r2;
  bool ok;
};

// attempt at ring offset `at` (word index inside the stream's ring, < ring_words)
__device__ __forceinline__ Attempt load_attempt(const uint32_t *__restrict__ ring, size_t ring_words, size_t at) {
#pragma clang fp contract(off)
  uint32_t w[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    size_t idx = at + j;
    if (idx >= ring_words) idx -= ring_words;
    w[j] = mt_temper(ring[idx]);
  }
  const double d1 = (static_cast<double>(w[0] >> 5) * 67108864.0 + static_cast<double>(w[1] >> 6)) / 9007199254740992.0;
  const double d2 = (static_cast<double>(w[2] >> 5) * 67108864.0 + static_cast<double>(w[3] >> 6)) / 9007199254740992.0;
  Attempt a;
  a.x1 = 2.0 * d1 - 1.0;
  a.x2 = 2.0 * d2 - 1.0;
  const double p1 = a.x1 * a.x1, p2 = a.x2 * a.x2;
  a.r2 = p1 + p2;
  a.ok = !(a.r2 >= 1.0 || a.r2 == 0.0);
  return a;
}

// order of the attempts of a tile: k = tile * 1024 + q * 256 + thread, q = 0..3
__global__ __launch_bounds__(kTileThreads) void polar_count(const uint32_t *__restrict__ rings, size_t ring_words,
                                                            const RngDyn *__restrict__ dyn, unsigned long long window,
                                                            unsigned ntiles, unsigned *__restrict__ tile_counts) {
  __shared__ unsigned s_cnt[kTileThreads / 64];
  const unsigned b = blockIdx.y;
  const uint32_t *ring = rings + static_cast<size_t>(b) * ring_words;
  const size_t base = static_cast<size_t>(dyn[b].pos % ring_words);
  unsigned cnt = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned long long k = static_cast<unsigned long long>(blockIdx.x) * kTileAttempts + q * kTileThreads + threadIdx.x;
    bool ok = false;
    if (k < window) {
      size_t at = base + 4 * k;
      if (at >= ring_words) at -= ring_words;
      ok = load_attempt(ring, ring_words, at).ok;
    }
    cnt += __popcll(__ballot(ok));
  }
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) tile_counts[static_cast<size_t>(b) * ntiles + blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// exclusive scan of a stream's tile counts (in place); the cached value of the previous draw goes
// out first; the state a draw without new attempts leaves behind
__global__ __launch_bounds__(1024) void polar_scan(unsigned *__restrict__ tile_counts, unsigned ntiles,
                                                   const RngDyn *__restrict__ dyn_in, RngDyn *__restrict__ dyn_out,
                                                   unsigned long long count, double *__restrict__ out,
                                                   int *__restrict__ err_flag) {
  __shared__ unsigned s_wave[16];
  __shared__ unsigned s_carry;
  const unsigned b = blockIdx.x;
  unsigned *cnt = tile_counts + static_cast<size_t>(b) * ntiles;
  const RngDyn in = dyn_in[b];
  const unsigned long long from_cache = (in.has_gauss && count > 0) ? 1 : 0;
  const unsigned long long need = (count - from_cache + 1) / 2;  // pairs
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (unsigned start = 0; start < ntiles; start += 1024) {
    const unsigned idx = start + threadIdx.x;
    const unsigned v = idx < ntiles ? cnt[idx] : 0;
    unsigned incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned up = __shfl_up(incl, d);
      if ((threadIdx.x & 63) >= d) incl += up;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    unsigned before = s_carry;
    for (unsigned w = 0; w < (threadIdx.x >> 6); ++w) before += s_wave[w];
    if (idx < ntiles) cnt[idx] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    RngDyn o = in;
    if (from_cache) {
      out[static_cast<size_t>(b) * count] = in.gauss;
      o.has_gauss = 0;
      o.gauss = 0.0;
    }
    if (need > s_carry) {  // window too short: the values of this draw are incomplete
      o.err = 1;
      *err_flag = 1;  // (pinned host word: psh_rng_check() reads it without touching the device)
    }
    dyn_out[b] = o;  // polar_write's closing attempt overwrites pos / gauss when need > 0
  }
}

__global__ __launch_bounds__(kTileThreads) void polar_write(const uint32_t *__restrict__ rings, size_t ring_words,
                                                            const RngDyn *__restrict__ dyn_in, RngDyn *__restrict__ dyn_out,
                                                            unsigned long long window, unsigned ntiles,
                                                            const unsigned *__restrict__ tile_offsets,
                                                            unsigned long long count, double *__restrict__ out) {
#pragma clang fp contract(off)
  __shared__ unsigned s_cnt[4][kTileThreads / 64];
  const unsigned b = blockIdx.y;
  const RngDyn in = dyn_in[b];
  const unsigned long long from_cache = (in.has_gauss && count > 0) ? 1 : 0;
  const unsigned long long need = (count - from_cache + 1) / 2;
  const unsigned long long first = tile_offsets[static_cast<size_t>(b) * ntiles + blockIdx.x];
  if (first >= need) return;  // uniform per workgroup
  const uint32_t *ring = rings + static_cast<size_t>(b) * ring_words;
  const size_t base = static_cast<size_t>(in.pos % ring_words);
  Attempt a[4];
  unsigned rank[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned long long k = static_cast<unsigned long long>(blockIdx.x) * kTileAttempts + q * kTileThreads + threadIdx.x;
    a[q].ok = false;
    if (k < window) {
      size_t at = base + 4 * k;
      if (at >= ring_words) at -= ring_words;
      a[q] = load_attempt(ring, ring_words, at);
    }
    const unsigned long long bal = __ballot(a[q].ok);
    rank[q] = __popcll(bal & ((1ull << (threadIdx.x & 63)) - 1ull));
    if ((threadIdx.x & 63) == 0) s_cnt[q][threadIdx.x >> 6] = __popcll(bal);
  }
  __syncthreads();
  double *dst = out + static_cast<size_t>(b) * count + from_cache;
  unsigned before = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int w = 0; w < kTileThreads / 64; ++w) {
      if (w == static_cast<int>(threadIdx.x >> 6)) {
        const unsigned long long p = first + before + rank[q];
        if (a[q].ok && p < need) {
          const double lg = crlog::log_cr(a[q].r2, c_log_table);
          const double f = sqrt(-2.0 * lg / a[q].r2);
          const double g2 = f * a[q].x2, g1 = f * a[q].x1;
          dst[2 * p] = g2;
          const bool spill = 2 * p + 1 >= count - from_cache;  // odd draw: the pair's second value is cached
          if (!spill) dst[2 * p + 1] = g1;
          if (p == need - 1) {
            const unsigned long long k = static_cast<unsigned long long>(blockIdx.x) * kTileAttempts + q * kTileThreads + threadIdx.x;
            dyn_out[b].pos = in.pos + 4 * (k + 1);
            dyn_out[b].gauss = spill ? g1 : 0.0;
            dyn_out[b].has_gauss = spill ? 1 : 0;
          }
        }
      }
      before += s_cnt[q][w];
    }
  }
}

// RandomState.uniform: value t of a draw reads words pos + 2 t, pos + 2 t + 1
__global__ __launch_bounds__(kTileThreads) void uniform_write(const uint32_t *__restrict__ rings, size_t ring_words,
                                                              const RngDyn *__restrict__ dyn_in, RngDyn *__restrict__ dyn_out,
                                                              unsigned long long count, double low, double range,
                                                              double *__restrict__ out) {
#pragma clang fp contract(off)
  const unsigned b = blockIdx.y;
  const RngDyn in = dyn_in[b];
  const uint32_t *ring = rings + static_cast<size_t>(b) * ring_words;
  const size_t base = static_cast<size_t>(in.pos % ring_words);
  double *dst = out + static_cast<size_t>(b) * count;
  const unsigned long long stride = static_cast<unsigned long long>(gridDim.x) * kTileThreads;
  for (unsigned long long t = static_cast<unsigned long long>(blockIdx.x) * kTileThreads + threadIdx.x; t < count; t += stride) {
    size_t at = base + static_cast<size_t>((2 * t) % ring_words);
    if (at >= ring_words) at -= ring_words;
    const size_t at1 = at + 1 >= ring_words ? at + 1 - ring_words : at + 1;
    const uint32_t w0 = mt_temper(ring[at]), w1 = mt_temper(ring[at1]);
    const double d = (static_cast<double>(w0 >> 5) * 67108864.0 + static_cast<double>(w1 >> 6)) / 9007199254740992.0;
    dst[t] = low + range * d;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    RngDyn o = in;
    o.pos = in.pos + 2 * count;
    dyn_out[b] = o;
  }
}



struct Rng {
  int streams = 0;
  size_t max_draw = 0;
  unsigned ring_blocks = 0;
  size_t ring_words = 0;
  uint32_t *rings = nullptr;
  RngDyn *dyn = nullptr;  // [2][streams], parity flips per draw
  unsigned *tiles = nullptr;
  unsigned max_tiles = 0;
  int parity = 0;
  unsigned long long produced_blocks = 1;  // blocks [0, produced_blocks) exist (block 0 = the initial key)
  // bounds of the streams' positions (absolute word index): what the host knows without reading
  // the device state - every pair takes 1 / (pi/4) attempts on average, 10 sigma either way per
  // draw; resync() collapses them onto the true positions when they have drifted too far apart
  double lo = 0.0, hi = 0.0;
  std::vector<uint32_t> key0;
  std::vector<int> pos0, has0;
  std::vector<double> gauss0;
  bool drawn = false;
  uint32_t *bases = nullptr;  // (streams, n_chunks, 624) start states of the chunks, or nullptr
  unsigned n_chunks = 0;      // chunks with a start state (0: one workgroup per stream produces in sequence)
  unsigned long long anchor_block = 0;  // block whose state is bases[.][0]: chunk c = blocks anchor + c C + 1 ...
  hipStream_t stream = nullptr;  // own stream: draws can run beside the main stream
  hipEvent_t ready = nullptr, fence = nullptr;
  bool on_side = false;  // the last draw ran on `stream` and has not been joined yet
  int *err_flag = nullptr;  // pinned host word the scan kernel raises when a stream's window ran short
};

unsigned long long window_for(unsigned long long pairs) {
  const double mean = static_cast<double>(pairs) / kAccept;
  return static_cast<unsigned long long>(mean * (1.0 + 1.0 / 1024.0)) + 8192;
}

// start states of chunks 1 .. n_chunks-1 from chunk 0's by doubling: chunk j = chunk j - 2^l jumped by D 2^l words
void rng_build_tree(Rng *r, hipStream_t stream) {
  for (int level = 0; (1u << level) < r->n_chunks; ++level) {
    const unsigned first = 1u << level, count = std::min(first, r->n_chunks - first);
    hipLaunchKernelGGL(mt_jump, dim3(count, r->streams), dim3(kProduceThreads), 0, stream, r->bases, r->n_chunks, level, first, count);
  }
}

// waits for the draws queued so far and reads the streams' true positions
int rng_resync(Rng *r) {
  PSH_HIP(hipStreamSynchronize(r->stream));
  PSH_HIP(hipStreamSynchronize(ctx().stream));
  std::vector<RngDyn> d(r->streams);
  PSH_HIP(hipMemcpy(d.data(), r->dyn + static_cast<size_t>(r->parity) * r->streams, d.size() * sizeof(RngDyn),
                    hipMemcpyDeviceToHost));
  unsigned long long lo = ~0ull, hi = 0;
  for (const RngDyn &s : d) {
    if (s.err) return fail(PSH_EHIP, "rng: a stream ran out of accepted attempts inside its window (a > 10 sigma event)");
    lo = std::min(lo, s.pos);
    hi = std::max(hi, s.pos);
  }
  r->lo = static_cast<double>(lo);
  r->hi = static_cast<double>(hi);
  return PSH_OK;
}

void rng_free(Rng *r) {
  if (!r) return;
  if (r->stream) (void)hipStreamSynchronize(r->stream);
  if (r->rings) (void)hipFree(r->rings);
  if (r->dyn) (void)hipFree(r->dyn);
  if (r->tiles) (void)hipFree(r->tiles);
  if (r->bases) (void)hipFree(r->bases);
  if (r->err_flag) (void)hipHostFree(r->err_flag);
  if (r->ready) (void)hipEventDestroy(r->ready);
  if (r->fence) (void)hipEventDestroy(r->fence);
  if (r->stream) (void)hipStreamDestroy(r->stream);
  delete r;
}

}  // namespace
}  // namespace psh

using psh::fail;

extern "C" int psh_rng_create(int n_streams, const uint32_t *keys_host, const int *pos_host, const int *has_gauss_host,
                              const double *gauss_host, size_t max_draw, int n_draws_hint, void **handle_out) {
  PSH_REQUIRE_INIT();
  if (!handle_out || !keys_host || !pos_host) return fail(PSH_EINVAL, "rng_create: NULL pointer");
  if (n_streams < 1 || n_streams > 4096) return fail(PSH_EINVAL, "rng_create: 1..4096 streams");
  if (max_draw == 0 || max_draw > (size_t(1) << 31)) return fail(PSH_EINVAL, "rng_create: 1..2^31 values per draw");
  for (int b = 0; b < n_streams; ++b)
    if (pos_host[b] < 0 || pos_host[b] > psh::kMtN) return fail(PSH_EINVAL, "rng_create: MT19937 position outside 0..624");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::Rng *r = new psh::Rng;
  r->streams = n_streams;
  r->max_draw = max_draw;
  const unsigned long long window = psh::window_for((max_draw + 1) / 2);
  // two windows + a block of slack on either side: the ring holds the unread tail of one draw and
  // the whole window of the next
  // chunked production rounds up to whole chunks: one chunk of slack
  r->ring_blocks = static_cast<unsigned>((2 * 4 * window) / psh::kMtN + 8 + (n_draws_hint > 0 ? psh::kChunkBlocks : 0));
  r->ring_words = static_cast<size_t>(r->ring_blocks) * psh::kMtN;
  r->max_tiles = static_cast<unsigned>((window + psh::kTileAttempts - 1) / psh::kTileAttempts);
  r->key0.assign(keys_host, keys_host + static_cast<size_t>(n_streams) * psh::kMtN);
  r->pos0.assign(pos_host, pos_host + n_streams);
  r->has0.assign(n_streams, 0);
  r->gauss0.assign(n_streams, 0.0);
  r->lo = *std::min_element(r->pos0.begin(), r->pos0.end());
  r->hi = *std::max_element(r->pos0.begin(), r->pos0.end());
  auto setup = [&]() -> int {
    PSH_HIP(hipMalloc(reinterpret_cast<void **>(&r->rings), static_cast<size_t>(n_streams) * r->ring_words * 4));
    PSH_HIP(hipMalloc(reinterpret_cast<void **>(&r->dyn), 2 * static_cast<size_t>(n_streams) * sizeof(psh::RngDyn)));
    PSH_HIP(hipMalloc(reinterpret_cast<void **>(&r->tiles), static_cast<size_t>(n_streams) * r->max_tiles * 4));
    PSH_HIP(hipHostMalloc(reinterpret_cast<void **>(&r->err_flag), sizeof(int), hipHostMallocDefault));
    *r->err_flag = 0;
    PSH_HIP(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking));
    PSH_HIP(hipEventCreateWithFlags(&r->ready, hipEventDisableTiming));
    PSH_HIP(hipEventCreateWithFlags(&r->fence, hipEventDisableTiming));
    std::vector<psh::RngDyn> d(2 * static_cast<size_t>(n_streams));
    for (int b = 0; b < n_streams; ++b) {
      psh::RngDyn &s = d[b];
      s.pos = static_cast<unsigned long long>(pos_host[b]);
      s.has_gauss = has_gauss_host ? (has_gauss_host[b] != 0) : 0;
      s.gauss = (s.has_gauss && gauss_host) ? gauss_host[b] : 0.0;
      s.err = 0;
      r->has0[b] = s.has_gauss;
      r->gauss0[b] = s.gauss;
      d[n_streams + b] = s;
      PSH_HIP(hipMemcpyAsync(r->rings + static_cast<size_t>(b) * r->ring_words, keys_host + static_cast<size_t>(b) * psh::kMtN,
                             psh::kMtN * 4, hipMemcpyHostToDevice, c.stream));
    }
    PSH_HIP(hipMemcpyAsync(r->dyn, d.data(), d.size() * sizeof(psh::RngDyn), hipMemcpyHostToDevice, c.stream));
    if (n_draws_hint > 0) {
      // start states of the chunks the hinted draws will read, by doubling (mt_jump): chunk j = chunk
      // j - 2^l jumped by D 2^l words, l = 0, 1, ...; on the handle's stream, the first draw waits for it
      const double words = static_cast<double>(n_draws_hint) * 4.0 * static_cast<double>(window) * 1.01 + 2.0 * psh::kMtN;
      const double chunk_words = static_cast<double>(psh::kMtN) * psh::kChunkBlocks;
      // (at most 2^12 chunks at a time: when they are used up the grid is anchored anew at the last block
      // produced, 3 ms of jumps on the producing stream)
      unsigned chunks = static_cast<unsigned>(std::min(words / chunk_words + 2.0, 4096.0));
      r->n_chunks = chunks;
      PSH_HIP(hipMalloc(reinterpret_cast<void **>(&r->bases), static_cast<size_t>(n_streams) * chunks * psh::kMtN * 4));
      for (int b = 0; b < n_streams; ++b)
        PSH_HIP(hipMemcpyAsync(r->bases + static_cast<size_t>(b) * chunks * psh::kMtN, keys_host + static_cast<size_t>(b) * psh::kMtN,
                               psh::kMtN * 4, hipMemcpyHostToDevice, c.stream));
      PSH_HIP(hipStreamSynchronize(c.stream));
      psh::rng_build_tree(r, r->stream);
      PSH_HIP(hipGetLastError());
      PSH_HIP(hipEventRecord(r->ready, r->stream));
      r->on_side = true;  // whoever draws first (on either stream) waits for the start states
    }
    PSH_HIP(hipStreamSynchronize(c.stream));  // the host vectors die here
    return PSH_OK;
  };
  const int rc = setup();
  if (rc != PSH_OK) {
    psh::rng_free(r);
    return rc;
  }
  *handle_out = r;
  return PSH_OK;
}

// Everything a draw needs before its own kernels: the stream it runs on (the library stream, or the
// handle's own one behind everything queued so far - psh_rng_wait() joins it) and the raw words up to
// `words_max` beyond the furthest position a stream can be at (lock held)
static int rng_prepare(psh::Rng *r, int side, double words_max, hipStream_t *stream_out) {
  psh::Context &c = psh::ctx();
  // which stream: the library stream, or the handle's own one behind everything queued so far
  // (whoever used out_dev before) - psh_rng_wait() joins it
  hipStream_t s = c.stream;
  if (r->on_side && !side) {  // an unjoined draw is still running on the handle's stream
    PSH_HIP(hipStreamWaitEvent(c.stream, r->ready, 0));
    r->on_side = false;
  }
  if (side) {
    PSH_HIP(hipEventRecord(r->fence, c.stream));
    PSH_HIP(hipStreamWaitEvent(r->stream, r->fence, 0));
    s = r->stream;
  }
  // words this draw may read: up to hi + words_max; the slot of block k is the slot of block
  // k - ring_blocks, which has to lie below every stream's position
  auto blocks_wanted = [&]() { return static_cast<unsigned long long>((r->hi + words_max) / psh::kMtN) + 2; };
  auto fits = [&](unsigned long long want) {
    // (one block of slack: get_state reads the block that holds word pos - 1)
    return want <= r->ring_blocks || static_cast<double>(want - r->ring_blocks + 1) * psh::kMtN <= r->lo;
  };
  unsigned long long want_blocks = blocks_wanted();
  if (want_blocks > r->produced_blocks && !fits(want_blocks)) {
    if (int rc = psh::rng_resync(r)) return rc;  // the bounds have drifted apart: read the positions
    want_blocks = blocks_wanted();
    if (!fits(want_blocks)) return fail(PSH_EUNSUPPORTED, "rng_randn: word ring too small for this sequence of draws");
  }
  // whole chunks from their own start states, each by its own workgroup, while start states last
  // (produced_blocks - 1 stays a multiple of the chunk length on this path)
  while (want_blocks > r->produced_blocks && r->bases) {
    unsigned long long rel = r->produced_blocks - 1 - r->anchor_block;
    if (rel % psh::kChunkBlocks != 0 || rel / psh::kChunkBlocks >= r->n_chunks) {
      // start states used up (or the sequential producer ran in between): anchor the chunk grid at the
      // last block produced and jump again - 3 ms on the producing stream
      r->anchor_block = r->produced_blocks - 1;
      hipLaunchKernelGGL(psh::mt_anchor, dim3(r->streams), dim3(psh::kProduceThreads), 0, s, r->rings, r->ring_blocks,
                         r->anchor_block, r->bases, r->n_chunks);
      psh::rng_build_tree(r, s);
      rel = 0;
    }
    const unsigned long long c0 = rel / psh::kChunkBlocks;
    unsigned long long c1 = (want_blocks - 1 - r->anchor_block + psh::kChunkBlocks - 1) / psh::kChunkBlocks;  // exclusive
    c1 = std::min<unsigned long long>(c1, r->n_chunks);
    if (c1 <= c0 || !fits(r->anchor_block + c1 * psh::kChunkBlocks + 1)) break;
    hipLaunchKernelGGL(psh::mt_produce_chunks, dim3(static_cast<unsigned>(c1 - c0), r->streams), dim3(psh::kProduceThreads), 0, s,
                       r->rings, r->ring_blocks, r->bases, r->n_chunks, static_cast<unsigned>(c0), r->anchor_block);
    r->produced_blocks = r->anchor_block + c1 * psh::kChunkBlocks + 1;
  }
  if (want_blocks > r->produced_blocks) {
    const unsigned long long n = want_blocks - r->produced_blocks;
    unsigned long long first = r->produced_blocks;
    unsigned long long left = n;
    while (left) {  // int argument: chunks of 2^30 blocks
      const int chunk = static_cast<int>(std::min<unsigned long long>(left, 1ull << 30));
      hipLaunchKernelGGL(psh::mt_produce, dim3(r->streams), dim3(psh::kProduceThreads), 0, s, r->rings, r->ring_blocks, first, chunk);
      first += chunk;
      left -= chunk;
    }
    r->produced_blocks = want_blocks;
  }
  *stream_out = s;
  return PSH_OK;
}

extern "C" int psh_rng_randn_dev(void *handle, size_t count, double *out_dev, int side) {
  PSH_REQUIRE_INIT();
  psh::Rng *r = static_cast<psh::Rng *>(handle);
  if (!r || (!out_dev && count)) return fail(PSH_EINVAL, "rng_randn: NULL pointer");
  if (count > r->max_draw) return fail(PSH_EINVAL, "rng_randn: %zu values per stream, the handle was made for %zu", count, r->max_draw);
  if (count == 0) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const unsigned long long pairs_max = (count + 1) / 2;  // no cached value
  const unsigned long long pairs_min = count / 2;        // every stream has one
  const unsigned long long window = psh::window_for(pairs_max);
  const unsigned ntiles = static_cast<unsigned>((window + psh::kTileAttempts - 1) / psh::kTileAttempts);
  hipStream_t s = nullptr;
  if (int rc = rng_prepare(r, side, 4.0 * static_cast<double>(window), &s)) return rc;
  const psh::RngDyn *din = r->dyn + static_cast<size_t>(r->parity) * r->streams;
  psh::RngDyn *dout = r->dyn + static_cast<size_t>(r->parity ^ 1) * r->streams;
  hipLaunchKernelGGL(psh::polar_count, dim3(ntiles, r->streams), dim3(psh::kTileThreads), 0, s, r->rings, r->ring_words, din,
                     window, ntiles, r->tiles);
  hipLaunchKernelGGL(psh::polar_scan, dim3(r->streams), dim3(1024), 0, s, r->tiles, ntiles, din, dout,
                     static_cast<unsigned long long>(count), out_dev, r->err_flag);
  hipLaunchKernelGGL(psh::polar_write, dim3(ntiles, r->streams), dim3(psh::kTileThreads), 0, s, r->rings, r->ring_words, din, dout,
                     window, ntiles, r->tiles, static_cast<unsigned long long>(count), out_dev);
  PSH_HIP(hipGetLastError());
  r->parity ^= 1;
  r->drawn = true;
  // attempts per pair: geometric, mean 1 / p, variance (1 - p) / p^2; 10 sigma either way
  const double p = psh::kAccept;
  auto attempts = [&](unsigned long long pairs, double sign) {
    const double mean = pairs / p, sd = std::sqrt(pairs * (1.0 - p)) / p;
    return std::max(static_cast<double>(pairs), mean + sign * (10.0 * sd + 16.0));
  };
  r->hi += 4.0 * std::min(static_cast<double>(window), attempts(pairs_max, 1.0));
  r->lo += 4.0 * attempts(pairs_min, -1.0);
  if (side) {
    PSH_HIP(hipEventRecord(r->ready, r->stream));
    r->on_side = true;
  }
  return PSH_OK;
}

// `count` values of RandomState.uniform(low, high) per stream (numpy/random/src/distributions/distributions.c
// random_uniform: low + (high - low) * next_double, two words per value, no rejection: the streams advance by
// exactly 2 count words; a cached normal value stays cached)
extern "C" int psh_rng_uniform_dev(void *handle, size_t count, double low, double high, double *out_dev, int side) {
  PSH_REQUIRE_INIT();
  psh::Rng *r = static_cast<psh::Rng *>(handle);
  if (!r || (!out_dev && count)) return fail(PSH_EINVAL, "rng_uniform: NULL pointer");
  if (count > r->max_draw) return fail(PSH_EINVAL, "rng_uniform: %zu values per stream, the handle was made for %zu", count, r->max_draw);
  if (count == 0) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipStream_t s = nullptr;
  const double words = 2.0 * static_cast<double>(count);
  if (int rc = rng_prepare(r, side, words, &s)) return rc;
  const psh::RngDyn *din = r->dyn + static_cast<size_t>(r->parity) * r->streams;
  psh::RngDyn *dout = r->dyn + static_cast<size_t>(r->parity ^ 1) * r->streams;
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>((count + psh::kTileThreads - 1) / psh::kTileThreads, 2048));
  hipLaunchKernelGGL(psh::uniform_write, dim3(blocks, r->streams), dim3(psh::kTileThreads), 0, s, r->rings, r->ring_words, din, dout,
                     static_cast<unsigned long long>(count), low, high - low, out_dev);
  PSH_HIP(hipGetLastError());
  r->parity ^= 1;
  r->drawn = true;
  r->hi += words;
  r->lo += words;
  if (side) {
    PSH_HIP(hipEventRecord(r->ready, r->stream));
    r->on_side = true;
  }
  return PSH_OK;
}

extern "C" int psh_rng_wait(void *handle) {
  PSH_REQUIRE_INIT();
  psh::Rng *r = static_cast<psh::Rng *>(handle);
  if (!r) return fail(PSH_EINVAL, "rng_wait: NULL handle");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  if (r->on_side) {
    PSH_HIP(hipSetDevice(c.device));
    PSH_HIP(hipStreamWaitEvent(c.stream, r->ready, 0));
    r->on_side = false;
  }
  return PSH_OK;
}

extern "C" int psh_rng_check(void *handle) {
  PSH_REQUIRE_INIT();
  psh::Rng *r = static_cast<psh::Rng *>(handle);
  if (!r) return fail(PSH_EINVAL, "rng_check: NULL handle");
  if (*static_cast<volatile int *>(r->err_flag))
    return fail(PSH_EHIP, "rng: a stream ran out of accepted attempts inside its window (a > 10 sigma event): the draw is incomplete");
  return PSH_OK;
}

extern "C" int psh_rng_get_state(void *handle, uint32_t *keys_host, int *pos_host, int *has_gauss_host, double *gauss_host) {
  PSH_REQUIRE_INIT();
  psh::Rng *r = static_cast<psh::Rng *>(handle);
  if (!r || !keys_host || !pos_host || !has_gauss_host || !gauss_host) return fail(PSH_EINVAL, "rng_get_state: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipStreamSynchronize(r->stream));
  PSH_HIP(hipStreamSynchronize(c.stream));
  std::vector<psh::RngDyn> d(r->streams);
  PSH_HIP(hipMemcpy(d.data(), r->dyn + static_cast<size_t>(r->parity) * r->streams, d.size() * sizeof(psh::RngDyn),
                    hipMemcpyDeviceToHost));
  for (int b = 0; b < r->streams; ++b) {
    if (d[b].err) return fail(PSH_EHIP, "rng: stream %d ran out of accepted attempts inside its window (a > 10 sigma event)", b);
    has_gauss_host[b] = d[b].has_gauss;
    gauss_host[b] = d[b].gauss;
    uint32_t *key = keys_host + static_cast<size_t>(b) * psh::kMtN;
    if (d[b].pos == static_cast<unsigned long long>(r->pos0[b])) {  // nothing read: the state as it came
      std::copy(r->key0.begin() + static_cast<size_t>(b) * psh::kMtN, r->key0.begin() + static_cast<size_t>(b + 1) * psh::kMtN, key);
      pos_host[b] = r->pos0[b];
      continue;
    }
    // NumPy twists lazily: after reading the last word of block k the state is (block k, 624)
    const unsigned long long blk = (d[b].pos - 1) / psh::kMtN;
    pos_host[b] = static_cast<int>(d[b].pos - blk * psh::kMtN);
    PSH_HIP(hipMemcpy(key, r->rings + static_cast<size_t>(b) * r->ring_words + static_cast<size_t>(blk % r->ring_blocks) * psh::kMtN,
                      psh::kMtN * 4, hipMemcpyDeviceToHost));
  }
  return PSH_OK;
}

extern "C" int psh_rng_destroy(void *handle) {
  PSH_REQUIRE_INIT();
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::rng_free(static_cast<psh::Rng *>(handle));
  return PSH_OK;
}
