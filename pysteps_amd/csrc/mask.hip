// The incremental precipitation mask of the STEPS member loop on the device:
// pysteps/nowcasts/utils.py:69-101, compute_dilated_mask(input_mask, kr, r) - called once per member
// and time step (nowcasts/steps.py:983,1210, sseps.py:472,821); in the reference 1 + r calls of
// scipy.ndimage.binary_dilation over the whole grid (52 ms at 1024^2 with the default r = 10: 43 % of
// what was left of a nowcasts.steps run once every other piece of the loop ran on the device).
//
//   mask0 = binary_dilation(input, kr)                          (:88)   generic structure, origin at its centre
//   mask  = mask0 + sum_{k=1..r} dilate^k(mask0, cross)          (:91-95)
//   out   = mask / mask.max()                                    (:98)
//
// Dilating k times by the 4-neighbour cross is the L1 ball of radius k (inside a rectangle the
// geodesic and the plain L1 distance agree; scipy's border value is 0, so nothing enters from
// outside), hence mask = max(0, r + 1 - d) with d the L1 distance to mask0 - a distance transform
// truncated at r + 1, separable: g = vertical distance to the nearest set pixel of the column,
// d = min_dj (|dj| + g(i, j + dj)).  Three element-wise kernels over bytes instead of r + 1 library
// passes; small integers and one exact division, so the result is bit-identical with the reference's
// (an empty mask gives 0 / 0 = NaN everywhere, like NumPy).
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace psh {
namespace {

constexpr int kThreads = 256;
constexpr int kGrid = 4096;

// The kernels take FOUR neighbouring pixels of a row per thread: a byte per lane and load (the first
// version) made the three sweeps cost 0.43 ms at 4096^2 for 0.2 GB of traffic - the vector memory
// pipeline spends the same time on a wave's 64 bytes as on its 256.  Unaligned 4-byte loads are plain
// global loads on gfx950; groups at the row ends and next to the image border go pixel by pixel.
__device__ __forceinline__ unsigned load4(const unsigned char *p) {
  unsigned w;
  __builtin_memcpy(&w, p, 4);
  return w;
}
// 0x01 in every byte of v that is not zero
__device__ __forceinline__ unsigned nonzero_bytes(unsigned v) {
  v = (v & 0x0f0f0f0fu) | ((v >> 4) & 0x0f0f0f0fu);
  v = (v & 0x03030303u) | ((v >> 2) & 0x03030303u);
  return (v | (v >> 1)) & 0x01010101u;
}

// out[p] = OR over the structure's offsets d of in[p - d] (scipy.ndimage.binary_dilation, origin 0,
// border_value 0); *any = 1 if anything is set
__global__ __launch_bounds__(kThreads) void mask_dilate(const unsigned char *__restrict__ in, int m, int n,
                                                        const short2 *__restrict__ taps, int ntaps, int reach,
                                                        unsigned char *__restrict__ out, int *any) {
  const int groups_per_row = (n + 3) >> 2;
  const size_t total = static_cast<size_t>(m) * groups_per_row, stride = static_cast<size_t>(gridDim.x) * kThreads;
  bool seen = false;
  for (size_t q = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; q < total; q += stride) {
    const int i = static_cast<int>(q / groups_per_row), j0 = static_cast<int>(q - static_cast<size_t>(i) * groups_per_row) << 2;
    const size_t p = static_cast<size_t>(i) * n + j0;
    if (j0 >= reach && j0 + 3 + reach < n) {  // every tap's four columns are inside the row
      unsigned v = 0;
      for (int t = 0; t < ntaps; ++t) {
        const int y = i - taps[t].x;
        if (y >= 0 && y < m) v |= load4(in + static_cast<size_t>(y) * n + (j0 - taps[t].y));
      }
      v = nonzero_bytes(v);
      __builtin_memcpy(out + p, &v, 4);
      seen |= v != 0;
    } else {
      for (int k = 0; k < 4 && j0 + k < n; ++k) {
        unsigned char v = 0;
        for (int t = 0; t < ntaps; ++t) {
          const int y = i - taps[t].x, x = j0 + k - taps[t].y;
          if (y >= 0 && y < m && x >= 0 && x < n && in[static_cast<size_t>(y) * n + x]) {
            v = 1;
            break;
          }
        }
        out[p + k] = v;
        seen |= v != 0;
      }
    }
  }
  if (__any(seen) && (threadIdx.x & 63) == 0) *any = 1;  // same value from every wave that saw one
}

// g = min(r + 1, distance to the nearest set pixel of the same column); mask0 holds 0 / 1
__global__ __launch_bounds__(kThreads) void mask_column_distance(const unsigned char *__restrict__ mask0, int m, int n,
                                                                 int r, unsigned char *__restrict__ g) {
  const int groups_per_row = (n + 3) >> 2;
  const size_t total = static_cast<size_t>(m) * groups_per_row, stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t q = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; q < total; q += stride) {
    const int i = static_cast<int>(q / groups_per_row), j0 = static_cast<int>(q - static_cast<size_t>(i) * groups_per_row) << 2;
    const size_t p = static_cast<size_t>(i) * n + j0;
    if (j0 + 3 < n) {
      // four byte lanes in one register: `open` = 0xff where no set pixel has been met yet
      const unsigned self = load4(mask0 + p);
      unsigned open = (self ^ 0x01010101u) * 0xffu;  // 0 / 1 bytes -> 0xff where the pixel is not set
      unsigned best = (static_cast<unsigned>(r + 1) * 0x01010101u) & open;
      for (int d = 1; d <= r && open; ++d) {
        unsigned hit = 0;
        if (i - d >= 0) hit |= load4(mask0 + p - static_cast<size_t>(d) * n);
        if (i + d < m) hit |= load4(mask0 + p + static_cast<size_t>(d) * n);
        const unsigned fresh = (hit * 0xffu) & open;  // 0 / 1 bytes: no carries between the lanes
        best = (best & ~fresh) | ((static_cast<unsigned>(d) * 0x01010101u) & fresh);
        open &= ~fresh;
      }
      __builtin_memcpy(g + p, &best, 4);
    } else {
      for (int k = 0; j0 + k < n; ++k) {
        int best = r + 1;
        if (mask0[p + k]) {
          best = 0;
        } else {
          for (int d = 1; d <= r; ++d) {
            const bool up = i - d >= 0 && mask0[p + k - static_cast<size_t>(d) * n];
            const bool down = i + d < m && mask0[p + k + static_cast<size_t>(d) * n];
            if (up || down) {
              best = d;
              break;
            }
          }
        }
        g[p + k] = static_cast<unsigned char>(best);
      }
    }
  }
}

// per-byte minimum of two words whose bytes are below 128
__device__ __forceinline__ unsigned min_bytes(unsigned a, unsigned b) {
  const unsigned ge = (((a | 0x80808080u) - b) >> 7) & 0x01010101u;  // 1 where a >= b (no borrow crosses a byte)
  const unsigned take_b = ge * 0xffu;
  return (b & take_b) | (a & ~take_b);
}

// d = min over the row of |dj| + g, out = max(0, r + 1 - d) / (r + 1 if anything is set, else 0)
__global__ __launch_bounds__(kThreads) void mask_rim(const unsigned char *__restrict__ g, int m, int n, int r,
                                                     const int *__restrict__ any, double *__restrict__ out) {
  const int groups_per_row = (n + 3) >> 2;
  const size_t total = static_cast<size_t>(m) * groups_per_row, stride = static_cast<size_t>(gridDim.x) * kThreads;
  const int cap = r + 1;
  const double top = *any ? static_cast<double>(cap) : 0.0;
  const bool packed = 2 * r + 1 < 128;  // |dj| + g stays below 128: four byte lanes in one register
  for (size_t q = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; q < total; q += stride) {
    const int i = static_cast<int>(q / groups_per_row), j0 = static_cast<int>(q - static_cast<size_t>(i) * groups_per_row) << 2;
    const size_t p = static_cast<size_t>(i) * n + j0;
    const int cnt = min(4, n - j0);
    int d[4];
    if (packed && j0 >= r && j0 + 3 + r < n) {
      // (a column |dj| away cannot bring less than |dj|: the walk outwards stops at the largest d of the four)
      unsigned d4 = load4(g + p);
      auto largest = [](unsigned v) { return max(max(v & 0xffu, (v >> 8) & 0xffu), max((v >> 16) & 0xffu, v >> 24)); };
      unsigned far = largest(d4);
      for (int dj = 1; dj <= r && static_cast<unsigned>(dj) < far; ++dj) {
        const unsigned add = static_cast<unsigned>(dj) * 0x01010101u;
        d4 = min_bytes(d4, load4(g + p - dj) + add);
        d4 = min_bytes(d4, load4(g + p + dj) + add);
        far = largest(d4);
      }
      for (int k = 0; k < 4; ++k) d[k] = static_cast<int>((d4 >> (8 * k)) & 0xffu);
    } else {
      for (int k = 0; k < cnt; ++k) {
        const int j = j0 + k;
        int dk = min(static_cast<int>(g[p + k]), cap);
        for (int dj = 1; dj <= r && dj < dk; ++dj) {
          if (j - dj >= 0) dk = min(dk, dj + static_cast<int>(g[p + k - dj]));
          if (j + dj < n) dk = min(dk, dj + static_cast<int>(g[p + k + dj]));
        }
        d[k] = dk;
      }
    }
    for (int k = 0; k < cnt; ++k) out[p + k] = static_cast<double>(cap - d[k]) / top;
  }
}

// ---- the same mask from a float64 field in two kernels, on BIT masks ------------------------------------------
// psh_steps_incremental_mask_dev: `precip_forecast >= precip_thr` (steps.py:1211) and compute_dilated_mask in one
// entry point.  The three byte kernels above move little (0.2 GB) but take 0.23 ms at 4096^2: a load per tap, per
// distance step and per pixel group.  Here:
//   wet_bits        one pass over the field -> one BIT per pixel (2 MiB at 4096^2: it stays in the L2s) and a flag
//                   "anything wet at all"
//   mask_from_bits  a WAVE owns a tile of 64 - 2 H columns and rows (H = r + the structure's reach): lane l holds the
//                   64-pixel word of image row y0 - H + l, columns x0 - H ... - the morphology of lk_open_bits:
//                   columns by shifts inside the word, rows by moving words between lanes.  mask0 = OR over the
//                   structure's taps, then r dilations by the cross, the level a pixel enters at recorded in bit
//                   planes (d = first k with the pixel in dilate^k(mask0); mask = r + 1 - d); what the shifts lose
//                   at the edge of the word / wave is the halo, consumed one ring per dilation.  Output: every
//                   lane turns its row's level planes into 64 - 2 H doubles... transposed back through LDS so that
//                   the stores are coalesced rows.
// Same small integers and the same one division as above: bit-identical.
constexpr int kBitsWaves = 4;  // waves (tiles) per workgroup
constexpr int kBitsMaxTaps = 96;  // set elements of the structure that travel as a kernel argument (no copy to queue)
struct BitTaps {
  short2 t[kBitsMaxTaps];
};

// `any`: a word that lives as long as the library; a launch that saw a wet pixel stores ITS generation number there
// (unique per call, never 0), the mask kernel behind it compares - no clearing launch in front of every call
__global__ __launch_bounds__(256) void wet_bits(const double *__restrict__ field, int m, int n, double thr,
                                                unsigned long long *__restrict__ bits, int words_per_row,
                                                int *__restrict__ any, int generation) {
  // one wave per 64-pixel word: a lane per pixel, the word by ballot
  const int lane = threadIdx.x & 63;
  const size_t wave = (static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x) >> 6;
  const size_t nwaves = (static_cast<size_t>(gridDim.x) * 256) >> 6;
  const size_t total = static_cast<size_t>(m) * words_per_row;
  bool seen = false;
  for (size_t w = wave; w < total; w += 4 * nwaves) {
    double v[4];
    size_t wi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      wi[k] = w + k * nwaves;
      const size_t wk = wi[k] < total ? wi[k] : w;
      const int row = static_cast<int>(wk / words_per_row), x = static_cast<int>(wk - static_cast<size_t>(row) * words_per_row) * 64 + lane;
      v[k] = x < n ? field[static_cast<size_t>(row) * n + x] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned long long word = __ballot(v[k] >= thr);  // NaN >= thr is false, like NumPy
      if (wi[k] < total) {
        if (lane == 0) bits[wi[k]] = word;
        seen |= word != 0ull;
      }
    }
  }
  if (seen && lane == 0) *any = generation;
}

__device__ __forceinline__ unsigned long long lane_word(unsigned long long v, int src_lane) {  // lane i gets lane src's word (0 outside the wave)
  const unsigned lo = static_cast<unsigned>(__shfl(static_cast<int>(v), src_lane & 63));
  const unsigned hi = static_cast<unsigned>(__shfl(static_cast<int>(v >> 32), src_lane & 63));
  const bool ok = src_lane >= 0 && src_lane < 64;
  return ok ? ((static_cast<unsigned long long>(hi) << 32) | lo) : 0ull;
}

template <int PLANES>
__global__ __launch_bounds__(64 * kBitsWaves) void mask_from_bits(const unsigned long long *__restrict__ bits, int words_per_row,
                                                                  int m, int n, const BitTaps taps, int ntaps,
                                                                  int halo, int r, const int *__restrict__ any, int generation,
                                                                  double *__restrict__ out, int tiles_x, int n_tiles) {
  __shared__ unsigned long long s_planes[kBitsWaves][PLANES + 1][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x * kBitsWaves + wave;
  if (tile >= n_tiles) return;  // (no barrier below: the waves of a workgroup are independent)
  const int side = 64 - 2 * halo;
  const int x0 = (tile % tiles_x) * side - halo, y0 = (tile / tiles_x) * side - halo;  // image position of bit 0 / lane 0
  // ---- the lane's word: 64 pixels of row y0 + lane from column x0 (outside the image: 0, scipy's border value)
  const int y = y0 + lane;
  unsigned long long in = 0ull;
  if (y >= 0 && y < m) {
    const unsigned long long *rowp = bits + static_cast<size_t>(y) * words_per_row;
    const int w0 = x0 >> 6, sh = x0 & 63;  // (arithmetic shift: x0 < 0 -> word -1)
    const unsigned long long a = (w0 >= 0 && w0 < words_per_row) ? rowp[w0] : 0ull;
    const unsigned long long b = (w0 + 1 >= 0 && w0 + 1 < words_per_row) ? rowp[w0 + 1] : 0ull;
    in = sh ? ((a >> sh) | (b << (64 - sh))) : a;
    // columns past the right edge of the image are not in the bit plane's tail (wet_bits writes 0 there)
  }
  // ---- mask0 = OR over the taps (dy, dx) of in[y - dy][x - dx]
  unsigned long long M = 0ull;
  for (int t = 0; t < ntaps; ++t) {
    const int dy = taps.t[t].x, dx = taps.t[t].y;
    const unsigned long long w = lane_word(in, lane - dy);
    M |= dx >= 0 ? (w << dx) : (w >> (-dx));
  }
  // (bits of mask0 that belong to pixels outside the image must not seed the rim: the reference dilates inside the
  // image only)
  {
    unsigned long long inside = 0ull;
    if (y >= 0 && y < m) {
      const int lo = max(0, -x0), hi = min(64, n - x0);  // bit range of image columns
      if (hi > lo) inside = (hi - lo == 64) ? ~0ull : (((1ull << (hi - lo)) - 1ull) << lo);
    }
    M &= inside;
    // ---- r dilations by the cross; D[b]: bit b of the level d a pixel enters at (d = 0: in mask0)
    unsigned long long D[PLANES];
#pragma unroll
    for (int b = 0; b < PLANES; ++b) D[b] = 0ull;
    unsigned long long cur = M;
    for (int k = 1; k <= r; ++k) {
      const unsigned long long up = lane_word(cur, lane - 1), dn = lane_word(cur, lane + 1);
      const unsigned long long nxt = (cur | (cur << 1) | (cur >> 1) | up | dn) & inside;
      const unsigned long long fresh = nxt & ~cur;
#pragma unroll
      for (int b = 0; b < PLANES; ++b)
        if ((k >> b) & 1) D[b] |= fresh;
      cur = nxt;
    }
    // pixels never reached: level r + 1 (mask value 0)
    const unsigned long long never = ~cur;
#pragma unroll
    for (int b = 0; b < PLANES; ++b)
      if (((r + 1) >> b) & 1) D[b] |= never;
    // ---- transpose through LDS: lane l now holds ROW l's planes; the stores want a lane per COLUMN
#pragma unroll
    for (int b = 0; b < PLANES; ++b) s_planes[wave][b][lane] = D[b];
  }
  __builtin_amdgcn_wave_barrier();
  const int cap = r + 1;
  const double top = *any == generation ? static_cast<double>(cap) : 0.0;
  const int x = x0 + lane;
  const bool col_ok = lane >= halo && lane < 64 - halo && x < n;
  for (int row = halo; row < 64 - halo; ++row) {
    const int yy = y0 + row;
    if (yy >= m) break;  // (uniform)
    int d = 0;
#pragma unroll
    for (int b = 0; b < PLANES; ++b) d |= static_cast<int>((s_planes[wave][b][row] >> lane) & 1ull) << b;
    if (col_ok) out[static_cast<size_t>(yy) * n + x] = static_cast<double>(cap - d) / top;
  }
}

}  // namespace
}  // namespace psh

extern "C" int psh_dilated_mask_dev(const unsigned char *mask_dev, int m, int n, const unsigned char *kr_host, int kh,
                                    int kw, int r, double *out_dev) {
  using namespace psh;
  PSH_REQUIRE_INIT();
  if (!mask_dev || !kr_host || !out_dev) return fail(PSH_EINVAL, "dilated_mask: NULL pointer");
  if (m <= 0 || n <= 0 || kh <= 0 || kw <= 0 || r < 0) return fail(PSH_EINVAL, "dilated_mask: invalid shape");
  if (r > 254) return fail(PSH_EUNSUPPORTED, "dilated_mask: at most 254 rim iterations");
  if (kh > 32767 || kw > 32767) return fail(PSH_EUNSUPPORTED, "dilated_mask: structuring element too large");
  Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  // offsets of the structure's set elements relative to its centre (scipy: size // 2), through the
  // pinned constant slot
  constexpr int kMaxTaps = static_cast<int>(kConstSlotFloats * sizeof(float) / sizeof(short2));
  float *slot_host = nullptr;
  const float *slot_dev = nullptr;
  if (int rc = const_slot(&slot_host, &slot_dev)) return rc;
  short2 *taps_host = reinterpret_cast<short2 *>(slot_host);
  int ntaps = 0;
  int reach = 0;  // largest column offset of a tap
  for (int y = 0; y < kh; ++y) {
    for (int x = 0; x < kw; ++x) {
      if (!kr_host[static_cast<size_t>(y) * kw + x]) continue;
      if (ntaps == kMaxTaps)
        return fail(PSH_EUNSUPPORTED, "dilated_mask: more than %d set elements in the structuring element", kMaxTaps);
      taps_host[ntaps].x = static_cast<short>(y - kh / 2);
      taps_host[ntaps].y = static_cast<short>(x - kw / 2);
      reach = std::max(reach, std::abs(x - kw / 2));
      ++ntaps;
    }
  }
  const size_t total = static_cast<size_t>(m) * n;
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, 256 + 2 * up(total))) return rc;  // [any | mask0 | g]
  char *base = static_cast<char *>(blk);
  int *any = reinterpret_cast<int *>(base);
  unsigned char *mask0 = reinterpret_cast<unsigned char *>(base + 256);
  unsigned char *g = mask0 + up(total);
  auto run = [&]() -> int {
    hipStream_t s = c.stream;
    if (ntaps)
      PSH_HIP(hipMemcpyAsync(const_cast<float *>(slot_dev), slot_host, static_cast<size_t>(ntaps) * sizeof(short2),
                             hipMemcpyHostToDevice, s));
    PSH_HIP(hipMemsetAsync(any, 0, sizeof(int), s));
    const size_t groups = static_cast<size_t>(m) * ((n + 3) / 4);  // four pixels of a row per thread
    const int grid = static_cast<int>(std::min<size_t>(kGrid, (groups + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(mask_dilate, dim3(grid), dim3(kThreads), 0, s, mask_dev, m, n,
                       reinterpret_cast<const short2 *>(slot_dev), ntaps, reach, mask0, any);
    hipLaunchKernelGGL(mask_column_distance, dim3(grid), dim3(kThreads), 0, s, mask0, m, n, r, g);
    hipLaunchKernelGGL(mask_rim, dim3(grid), dim3(kThreads), 0, s, g, m, n, r, any, out_dev);
    PSH_HIP(hipGetLastError());
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);  // stream-ordered
  return rc;
}

// `field >= threshold` (steps.py:1211) and compute_dilated_mask(., kr, r) (nowcasts/utils.py:69-101) of a float64 field in
// two kernels on bit masks (see wet_bits / mask_from_bits above).  Structures without their centre element, a halo
// (r + reach of the structure) above 24 pixels or r > 254: PSH_EUNSUPPORTED - the caller takes psh_ge_mask_dev +
// psh_dilated_mask_dev.
extern "C" int psh_steps_incremental_mask_dev(const double *field_dev, int m, int n, double threshold,
                                              const unsigned char *kr_host, int kh, int kw, int r, double *out_dev) {
  using namespace psh;
  PSH_REQUIRE_INIT();
  if (!field_dev || !kr_host || !out_dev) return fail(PSH_EINVAL, "incremental_mask: NULL pointer");
  if (m <= 0 || n <= 0 || kh <= 0 || kw <= 0 || r < 0) return fail(PSH_EINVAL, "incremental_mask: invalid shape");
  if (r > 254) return fail(PSH_EUNSUPPORTED, "incremental_mask: at most 254 rim iterations");
  Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  BitTaps taps;
  int ntaps = 0, reach = 0;
  bool centre = false;
  for (int y = 0; y < kh; ++y) {
    for (int x = 0; x < kw; ++x) {
      if (!kr_host[static_cast<size_t>(y) * kw + x]) continue;
      if (ntaps == kBitsMaxTaps) return fail(PSH_EUNSUPPORTED, "incremental_mask: more than %d set elements in the structure", kBitsMaxTaps);
      taps.t[ntaps].x = static_cast<short>(y - kh / 2);
      taps.t[ntaps].y = static_cast<short>(x - kw / 2);
      reach = std::max(reach, std::max(std::abs(x - kw / 2), std::abs(y - kh / 2)));
      centre |= (y == kh / 2 && x == kw / 2);
      ++ntaps;
    }
  }
  for (int t = ntaps; t < kBitsMaxTaps; ++t) taps.t[t] = make_short2(0, 0);
  // (with the centre element mask0 contains the input: "anything set in mask0" = "anything wet")
  if (!centre) return fail(PSH_EUNSUPPORTED, "incremental_mask: the structure lacks its centre element");
  const int halo = r + reach;
  if (halo > 24) return fail(PSH_EUNSUPPORTED, "incremental_mask: rim + structure reach %d pixels (at most 24)", halo);
  const int words_per_row = (n + 63) / 64;
  const size_t bit_bytes = static_cast<size_t>(m) * words_per_row * sizeof(unsigned long long);
  // the "anything wet" word and its generation counter (see wet_bits)
  if (!c.mask_any) {
    PSH_HIP(hipMalloc(reinterpret_cast<void **>(&c.mask_any), sizeof(int)));
    PSH_HIP(hipMemsetAsync(c.mask_any, 0, sizeof(int), c.stream));  // once, in stream order
    c.mask_generation = 0;
  }
  c.mask_generation = c.mask_generation == 0x7fffffff ? 1 : c.mask_generation + 1;
  int *any = c.mask_any;
  const int generation = c.mask_generation;
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, bit_bytes)) return rc;
  unsigned long long *bits = static_cast<unsigned long long *>(blk);
  auto run = [&]() -> int {
    hipStream_t s = c.stream;
    const size_t words = static_cast<size_t>(m) * words_per_row;
    const int wgrid = static_cast<int>(std::min<size_t>((words + 15) / 16, static_cast<size_t>(c.cu_count) * 16));
    hipLaunchKernelGGL(wet_bits, dim3(std::max(wgrid, 1)), dim3(256), 0, s, field_dev, m, n, threshold, bits, words_per_row, any,
                       generation);
    const int side = 64 - 2 * halo;
    const int tiles_x = (n + side - 1) / side, tiles_y = (m + side - 1) / side, n_tiles = tiles_x * tiles_y;
    const dim3 grid((n_tiles + kBitsWaves - 1) / kBitsWaves), block(64 * kBitsWaves);
    if (r + 1 < 16) {
      hipLaunchKernelGGL(mask_from_bits<4>, grid, block, 0, s, bits, words_per_row, m, n, taps, ntaps, halo, r, any, generation, out_dev,
                         tiles_x, n_tiles);
    } else {
      hipLaunchKernelGGL(mask_from_bits<8>, grid, block, 0, s, bits, words_per_row, m, n, taps, ntaps, halo, r, any, generation, out_dev,
                         tiles_x, n_tiles);
    }
    PSH_HIP(hipGetLastError());
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);  // stream-ordered
  return rc;
}
