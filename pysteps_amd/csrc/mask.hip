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
