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

#include "common.h"

namespace psh {
namespace {

constexpr int kThreads = 256;
constexpr int kGrid = 4096;

// out[p] = OR over the structure's offsets d of in[p - d] (scipy.ndimage.binary_dilation, origin 0,
// border_value 0); *any = 1 if anything is set
__global__ __launch_bounds__(kThreads) void mask_dilate(const unsigned char *__restrict__ in, int m, int n,
                                                        const short2 *__restrict__ taps, int ntaps,
                                                        unsigned char *__restrict__ out, int *any) {
  const size_t total = static_cast<size_t>(m) * n, stride = static_cast<size_t>(gridDim.x) * kThreads;
  bool seen = false;
  for (size_t p = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; p < total; p += stride) {
    const int i = static_cast<int>(p / n), j = static_cast<int>(p - static_cast<size_t>(i) * n);
    unsigned char v = 0;
    for (int t = 0; t < ntaps; ++t) {
      const int y = i - taps[t].x, x = j - taps[t].y;
      if (y >= 0 && y < m && x >= 0 && x < n && in[static_cast<size_t>(y) * n + x]) {
        v = 1;
        break;
      }
    }
    out[p] = v;
    seen |= v != 0;
  }
  if (__any(seen) && (threadIdx.x & 63) == 0) *any = 1;  // same value from every wave that saw one
}

// g = min(r + 1, distance to the nearest set pixel of the same column)
__global__ __launch_bounds__(kThreads) void mask_column_distance(const unsigned char *__restrict__ mask0, int m, int n,
                                                                 int r, unsigned char *__restrict__ g) {
  const size_t total = static_cast<size_t>(m) * n, stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t p = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; p < total; p += stride) {
    const int i = static_cast<int>(p / n);
    int best = r + 1;
    if (mask0[p]) {
      best = 0;
    } else {
      for (int d = 1; d <= r; ++d) {
        const bool up = i - d >= 0 && mask0[p - static_cast<size_t>(d) * n];
        const bool down = i + d < m && mask0[p + static_cast<size_t>(d) * n];
        if (up || down) {
          best = d;
          break;
        }
      }
    }
    g[p] = static_cast<unsigned char>(best);
  }
}

// d = min over the row of |dj| + g, out = max(0, r + 1 - d) / (r + 1 if anything is set, else 0)
__global__ __launch_bounds__(kThreads) void mask_rim(const unsigned char *__restrict__ g, int m, int n, int r,
                                                     const int *__restrict__ any, double *__restrict__ out) {
  const size_t total = static_cast<size_t>(m) * n, stride = static_cast<size_t>(gridDim.x) * kThreads;
  const int cap = r + 1;
  const double top = *any ? static_cast<double>(cap) : 0.0;
  for (size_t p = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; p < total; p += stride) {
    const int i = static_cast<int>(p / n), j = static_cast<int>(p - static_cast<size_t>(i) * n);
    int d = g[p] < cap ? g[p] : cap;
    for (int dj = 1; dj <= r && dj < d; ++dj) {  // a column |dj| away cannot bring less than |dj|
      if (j - dj >= 0) d = min(d, dj + static_cast<int>(g[p - dj]));
      if (j + dj < n) d = min(d, dj + static_cast<int>(g[p + dj]));
    }
    out[p] = static_cast<double>(cap - d) / top;
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
  for (int y = 0; y < kh; ++y) {
    for (int x = 0; x < kw; ++x) {
      if (!kr_host[static_cast<size_t>(y) * kw + x]) continue;
      if (ntaps == kMaxTaps)
        return fail(PSH_EUNSUPPORTED, "dilated_mask: more than %d set elements in the structuring element", kMaxTaps);
      taps_host[ntaps].x = static_cast<short>(y - kh / 2);
      taps_host[ntaps].y = static_cast<short>(x - kw / 2);
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
    const int grid = static_cast<int>(std::min<size_t>(kGrid, (total + kThreads - 1) / kThreads));
    hipLaunchKernelGGL(mask_dilate, dim3(grid), dim3(kThreads), 0, s, mask_dev, m, n,
                       reinterpret_cast<const short2 *>(slot_dev), ntaps, mask0, any);
    hipLaunchKernelGGL(mask_column_distance, dim3(grid), dim3(kThreads), 0, s, mask0, m, n, r, g);
    hipLaunchKernelGGL(mask_rim, dim3(grid), dim3(kThreads), 0, s, g, m, n, r, any, out_dev);
    PSH_HIP(hipGetLastError());
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);  // stream-ordered
  return rc;
}
