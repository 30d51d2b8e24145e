// Scale-space blob detection (the `fd_method="blob"` feature detector of dense Lucas-Kanade), for gfx950.
//
// pysteps/feature/blob.py:32-140 hands the image to scikit-image's blob_log / blob_dog [third party; 0.18.3 in this
// project's images: skimage/feature/blob.py, skimage/feature/peak.py], which are a thin layer over SciPy
// [scipy/ndimage/_filters.py, src/ni_filters.c]:
//   LoG   cube[k] = -gaussian_laplace(image, s_k) * s_k^2      gaussian_laplace = G''(rows) G(cols) + G(rows) G''(cols),
//   DoG   cube[k] = (G(s_k) - G(s_k+1)) image * s_k            every factor one correlate1d, mode "reflect", truncate 4
//   peaks cube == maximum_filter(cube, 3 x 3 x 3, mode "constant") and cube > threshold
// The reference spends its time in the 4 x K one-dimensional convolutions with up to 8 sigma + 1 taps over the whole
// grid and the three maximum-filter passes over the K-plane cube (minutes at 4096^2 on a core); the pruning of
// overlapping blobs works on the few hundred peaks and stays on the host (pysteps_amd/feature/blob.py).
//
// Arithmetic = SciPy's, operation by operation, FP contraction off:
//   correlate1d, symmetric kernel (ni_filters.c NI_Correlate1D): o = x[c] w[0]; for j = r .. 1: o += (x[c - j] + x[c + j]) w[j]
//   in double whatever the image's dtype; every pass stores in the image's dtype (float32 images round there);
//   "reflect" = (d c b a | a b c d | d c b a), repeated for kernels longer than the line.
//   maximum_filter1d (NI_MinOrMaxFilter1D: Harter's sliding-window ring of (value, death) pairs), size 3, along rows,
//   then columns, then scales, the line extended by one 0 on either side.  With NaNs in the cube (an image with
//   missing pixels gives NaN within 4 sigma of them) the ring's comparisons (`val >= front`, `back <= val`, both false
//   for NaN) decide what the filter returns next to them, and that depends on more than the three values under the
//   window - so the ring is run as SciPy runs it, one thread per line, instead of a 27-point maximum.
// The weights come from the caller (NumPy's exp, as scipy.ndimage._gaussian_kernel1d evaluates them).
//
// Layout: image (m, n) float32 or float64; cube (K, m, n) float64 planes (scikit-image stacks the scales last; only
// the peaks' coordinates leave this file).  None of this is on the headline path: the kernels read through the caches
// (a vertical pass reads 2 r + 1 rows per output row, all lanes of a wave the same rows) rather than staging tiles.
#include <algorithm>
#include <vector>

#include "common.h"

namespace psh {
namespace {

constexpr int kBlobThreads = 256;
constexpr int kBlobMaxRadius = 2048;

// index of line element i under scipy's "reflect" extension of a line of `len` elements
__device__ __forceinline__ int reflect_index(int i, int len) {
  const int period = 2 * len;
  i %= period;
  if (i < 0) i += period;
  return i < len ? i : period - 1 - i;
}

// correlate1d down the rows (axis 0) with two symmetric kernels at once (centre + left half each): the Gaussian and
// its second derivative read the same image.  64 x 4 pixels per workgroup.
template <typename T>
__global__ __launch_bounds__(kBlobThreads) void blob_corr_axis0(const T *__restrict__ in, int m, int n,
                                                                const double *__restrict__ wa, const double *__restrict__ wb,
                                                                int r, T *__restrict__ out_a, T *__restrict__ out_b) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= n || y >= m) return;
  const double c = static_cast<double>(in[static_cast<size_t>(y) * n + x]);
  double acc_a = c * wa[0], acc_b = wb ? c * wb[0] : 0.0;
  const bool inside = y - r >= 0 && y + r < m;
  for (int j = r; j >= 1; --j) {
    const int lo = inside ? y - j : reflect_index(y - j, m), hi = inside ? y + j : reflect_index(y + j, m);
    const double s = static_cast<double>(in[static_cast<size_t>(lo) * n + x]) + static_cast<double>(in[static_cast<size_t>(hi) * n + x]);
    const double pa = s * wa[j];
    acc_a = acc_a + pa;
    if (wb) {
      const double pb = s * wb[j];
      acc_b = acc_b + pb;
    }
  }
  out_a[static_cast<size_t>(y) * n + x] = static_cast<T>(acc_a);
  if (wb) out_b[static_cast<size_t>(y) * n + x] = static_cast<T>(acc_b);
}

// correlate1d along the rows' elements (axis 1) of one or two images, each with its own kernel; a workgroup stages
// its 256 outputs' span of the row in LDS.
//   LoG (in_b != nullptr): sum = T(corr(in_a, wa)) + T(corr(in_b, wb)) in T (gaussian_laplace: `output += tmp`),
//                          cube = double(-sum) * scale
//   DoG (in_b == nullptr): out_t = T(corr(in_a, wa))   (the smoothed image; the differences are taken by blob_dog_plane)
template <typename T>
__global__ __launch_bounds__(kBlobThreads) void blob_corr_axis1(const T *__restrict__ in_a, const T *__restrict__ in_b, int m, int n,
                                                                const double *__restrict__ wa, const double *__restrict__ wb,
                                                                int r, double scale, T *__restrict__ out_t,
                                                                double *__restrict__ cube_plane) {
#pragma clang fp contract(off)
  extern __shared__ double s_row[];  // [2][kBlobThreads + 2 r]
  const int y = blockIdx.y, x0 = blockIdx.x * kBlobThreads, span = kBlobThreads + 2 * r;
  double *row_a = s_row, *row_b = s_row + span;
  for (int i = threadIdx.x; i < span; i += kBlobThreads) {
    const int src = reflect_index(x0 - r + i, n);
    row_a[i] = static_cast<double>(in_a[static_cast<size_t>(y) * n + src]);
    if (in_b) row_b[i] = static_cast<double>(in_b[static_cast<size_t>(y) * n + src]);
  }
  __syncthreads();
  const int x = x0 + threadIdx.x;
  if (x >= n) return;
  const int c = threadIdx.x + r;
  double acc_a = row_a[c] * wa[0], acc_b = in_b ? row_b[c] * wb[0] : 0.0;
  for (int j = r; j >= 1; --j) {
    const double sa = row_a[c - j] + row_a[c + j];
    const double pa = sa * wa[j];
    acc_a = acc_a + pa;
    if (in_b) {
      const double sb = row_b[c - j] + row_b[c + j];
      const double pb = sb * wb[j];
      acc_b = acc_b + pb;
    }
  }
  const size_t at = static_cast<size_t>(y) * n + x;
  if (in_b) {
    const T l0 = static_cast<T>(acc_a), l1 = static_cast<T>(acc_b);
    const T sum = l0 + l1;
    const T neg = -sum;
    cube_plane[at] = static_cast<double>(neg) * scale;
  } else {
    out_t[at] = static_cast<T>(acc_a);
  }
}

// DoG plane: (G_k - G_k+1) in the image's dtype, times s_k
template <typename T>
__global__ __launch_bounds__(kBlobThreads) void blob_dog_plane(const T *__restrict__ g0, const T *__restrict__ g1, size_t count,
                                                               double scale, double *__restrict__ cube_plane) {
#pragma clang fp contract(off)
  const size_t stride = static_cast<size_t>(gridDim.x) * kBlobThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlobThreads + threadIdx.x; i < count; i += stride) {
    const T d = g0[i] - g1[i];
    cube_plane[i] = static_cast<double>(d) * scale;
  }
}

// NI_MinOrMaxFilter1D (maximum, filter size 3, mode "constant" with value 0) over lines of `len` elements `step`
// apart; one thread per line, `lines` of them, line l starting at base(l) = (l / inner) * outer_step + (l % inner).
// The ring holds at most three live (value, death) pairs: kept as a front-aligned queue in registers.
__global__ __launch_bounds__(kBlobThreads) void blob_max3_lines(const double *__restrict__ in, double *__restrict__ out, size_t lines,
                                                                size_t inner, size_t outer_step, int len, size_t step) {
  const size_t l = static_cast<size_t>(blockIdx.x) * kBlobThreads + threadIdx.x;
  if (l >= lines) return;
  const size_t base = (l / inner) * outer_step + (l % inner);
  const double *src = in + base;
  double *dst = out + base;
  constexpr int F = 3;
  double v0 = 0.0, v1 = 0.0, v2 = 0.0;  // queue, front first; v0 = the extension's 0 in front of the line
  int d0 = F, d1 = 0, d2 = 0, size = 1;
  for (int ll = 1; ll < F + len - 1; ++ll) {
    const double val = ll <= len ? src[static_cast<size_t>(ll - 1) * step] : 0.0;
    if (d0 == ll) {  // the front pair dies
      v0 = v1;
      d0 = d1;
      v1 = v2;
      d1 = d2;
      --size;
    }
    if (val >= v0) {  // (false if either is NaN)
      v0 = val;
      d0 = ll + F;
      size = 1;
    } else {
      // drop the pairs at the back that are <= val (a NaN at the back stays, and shields what is in front of it)
      if (size == 3 && v2 <= val) size = 2;
      if (size == 2 && v1 <= val) size = 1;
      if (size == 1) {
        v1 = val;
        d1 = ll + F;
      } else {
        v2 = val;
        d2 = ll + F;
      }
      ++size;
    }
    if (ll >= F - 1) dst[static_cast<size_t>(ll - (F - 1)) * step] = v0;
  }
}

// peak_local_max's mask: cube == its 3 x 3 x 3 maximum and cube > threshold -> (y, x, k) + value appended to a list;
// `not_max` counts the elements that differ from their neighbourhood maximum (0: the cube is "trivial", no peak at all)
__global__ __launch_bounds__(kBlobThreads) void blob_collect(const double *__restrict__ cube, const double *__restrict__ cmax, int K,
                                                             int m, int n, double threshold, int capacity,
                                                             int *__restrict__ coords, double *__restrict__ values,
                                                             unsigned *__restrict__ counters /* [0] peaks, [1] not_max */) {
  const size_t plane = static_cast<size_t>(m) * n, total = plane * K;
  const size_t stride = static_cast<size_t>(gridDim.x) * kBlobThreads;
  unsigned differ = 0;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlobThreads + threadIdx.x; i < total; i += stride) {
    const double v = cube[i];
    const bool is_max = v == cmax[i];
    differ += is_max ? 0u : 1u;
    if (is_max && v > threshold) {
      const unsigned at = atomicAdd(&counters[0], 1u);
      if (at < static_cast<unsigned>(capacity)) {
        const int k = static_cast<int>(i / plane);
        const size_t p = i - static_cast<size_t>(k) * plane;
        coords[3 * at + 0] = static_cast<int>(p / n);
        coords[3 * at + 1] = static_cast<int>(p % n);
        coords[3 * at + 2] = k;
        values[at] = v;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) differ += __shfl_xor(differ, d);
  if ((threadIdx.x & 63) == 0 && differ) atomicAdd(&counters[1], differ);
}

// values of a plane at listed pixels (the blob intensities pysteps/feature/blob.py:126-131 sorts by)
__global__ void blob_gather(const double *__restrict__ plane, int n, const int *__restrict__ yx, int count, double *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = plane[static_cast<size_t>(yx[2 * i]) * n + yx[2 * i + 1]];
}

template <typename T>
int blob_cube(const T *image, int m, int n, int method, const double *sigmas, int nsig, const int *radius, const double *weights,
              double *cube, hipStream_t stream) {
  const size_t plane = static_cast<size_t>(m) * n;
  size_t wtotal = 0;
  int rmax = 0;
  for (int k = 0; k < nsig; ++k) {
    wtotal += 2 * static_cast<size_t>(radius[k] + 1);
    rmax = std::max(rmax, radius[k]);
  }
  // [weights | tmp_a | tmp_b | g_prev | g_cur]
  const size_t wbytes = (wtotal * sizeof(double) + 255) & ~static_cast<size_t>(255);
  const size_t pbytes = (plane * sizeof(T) + 255) & ~static_cast<size_t>(255);
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, wbytes + 4 * pbytes)) return rc;
  char *base = static_cast<char *>(blk);
  double *w_dev = reinterpret_cast<double *>(base);
  T *tmp_a = reinterpret_cast<T *>(base + wbytes), *tmp_b = reinterpret_cast<T *>(base + wbytes + pbytes);
  T *g_prev = reinterpret_cast<T *>(base + wbytes + 2 * pbytes), *g_cur = reinterpret_cast<T *>(base + wbytes + 3 * pbytes);
  auto run = [&]() -> int {
    // (the caller's array may go once this call returns: the copy is waited for - a detection is not a launch chain
    // anybody queues behind)
    PSH_HIP(hipMemcpyAsync(w_dev, weights, wtotal * sizeof(double), hipMemcpyHostToDevice, stream));
    PSH_HIP(hipStreamSynchronize(stream));
    const dim3 grid0((n + 63) / 64, (m + 3) / 4), grid1((n + kBlobThreads - 1) / kBlobThreads, m);
    const size_t lds = 2 * static_cast<size_t>(kBlobThreads + 2 * rmax) * sizeof(double);
    if (lds > 64 * 1024) PSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&blob_corr_axis1<T>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    size_t woff = 0;
    for (int k = 0; k < nsig; ++k) {
      const int r = radius[k];
      const double *w_g = w_dev + woff, *w_d2 = w_g + (r + 1);
      woff += 2 * static_cast<size_t>(r + 1);
      const size_t lds_k = 2 * static_cast<size_t>(kBlobThreads + 2 * r) * sizeof(double);
      if (method == 0) {
        // rows: tmp_a = G'' image, tmp_b = G image; columns: G tmp_a + G'' tmp_b -> cube[k]
        hipLaunchKernelGGL(blob_corr_axis0<T>, grid0, dim3(kBlobThreads), 0, stream, image, m, n, w_d2, w_g, r, tmp_a, tmp_b);
        hipLaunchKernelGGL(blob_corr_axis1<T>, grid1, dim3(kBlobThreads), lds_k, stream, static_cast<const T *>(tmp_a),
                           static_cast<const T *>(tmp_b), m, n, w_g, w_d2, r, sigmas[k] * sigmas[k], static_cast<T *>(nullptr),
                           cube + static_cast<size_t>(k) * plane);
      } else {
        hipLaunchKernelGGL(blob_corr_axis0<T>, grid0, dim3(kBlobThreads), 0, stream, image, m, n, w_g, static_cast<const double *>(nullptr),
                           r, tmp_a, static_cast<T *>(nullptr));
        hipLaunchKernelGGL(blob_corr_axis1<T>, grid1, dim3(kBlobThreads), lds_k, stream, static_cast<const T *>(tmp_a),
                           static_cast<const T *>(nullptr), m, n, w_g, static_cast<const double *>(nullptr), r, 0.0, g_cur,
                           static_cast<double *>(nullptr));
        if (k > 0) {
          const unsigned grid = static_cast<unsigned>(std::min<size_t>((plane + kBlobThreads - 1) / kBlobThreads, 8192));
          hipLaunchKernelGGL(blob_dog_plane<T>, dim3(grid), dim3(kBlobThreads), 0, stream, static_cast<const T *>(g_prev),
                             static_cast<const T *>(g_cur), plane, sigmas[k - 1], cube + static_cast<size_t>(k - 1) * plane);
        }
        std::swap(g_prev, g_cur);
      }
    }
    PSH_HIP(hipGetLastError());
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);  // stream-ordered
  return rc;
}

}  // namespace
}  // namespace psh

using psh::fail;

extern "C" int psh_blob_cube_dev(const void *image_dev, int image_is_f32, int m, int n, int method, const double *sigmas_host,
                                 int nsig, const int *radius_host, const double *weights_host, double *cube_dev) {
  PSH_REQUIRE_INIT();
  if (!image_dev || !sigmas_host || !radius_host || !weights_host || !cube_dev) return fail(PSH_EINVAL, "blob_cube: NULL pointer");
  if (m <= 0 || n <= 0 || static_cast<size_t>(m) * n > (size_t(1) << 30)) return fail(PSH_EINVAL, "blob_cube: invalid shape (%d,%d)", m, n);
  if (method != 0 && method != 1) return fail(PSH_EINVAL, "blob_cube: method 0 (LoG) or 1 (DoG)");
  if (nsig < (method ? 2 : 1) || nsig > 64) return fail(PSH_EUNSUPPORTED, "blob_cube: %d scales (1..64; DoG needs two)", nsig);
  for (int k = 0; k < nsig; ++k)
    if (radius_host[k] < 0 || radius_host[k] > psh::kBlobMaxRadius)
      return fail(PSH_EUNSUPPORTED, "blob_cube: kernel radius %d (0..%d)", radius_host[k], psh::kBlobMaxRadius);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (image_is_f32)
    return psh::blob_cube(static_cast<const float *>(image_dev), m, n, method, sigmas_host, nsig, radius_host, weights_host, cube_dev, c.stream);
  return psh::blob_cube(static_cast<const double *>(image_dev), m, n, method, sigmas_host, nsig, radius_host, weights_host, cube_dev, c.stream);
}

extern "C" int psh_blob_peaks_dev(const double *cube_dev, int K, int m, int n, double threshold, int capacity, int *coords_host,
                                  double *values_host, int *count_host) {
  using namespace psh;
  PSH_REQUIRE_INIT();
  if (!cube_dev || !coords_host || !values_host || !count_host) return fail(PSH_EINVAL, "blob_peaks: NULL pointer");
  if (K < 1 || m <= 0 || n <= 0 || capacity < 1) return fail(PSH_EINVAL, "blob_peaks: invalid shape or capacity");
  Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * n, total = plane * K;
  const size_t cbytes = (total * sizeof(double) + 255) & ~static_cast<size_t>(255);
  const size_t lbytes = (static_cast<size_t>(capacity) * 3 * sizeof(int) + 255) & ~static_cast<size_t>(255);
  const size_t vbytes = (static_cast<size_t>(capacity) * sizeof(double) + 255) & ~static_cast<size_t>(255);
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, 256 + cbytes + lbytes + vbytes)) return rc;
  char *base = static_cast<char *>(blk);
  unsigned *counters = reinterpret_cast<unsigned *>(base);
  double *cmax = reinterpret_cast<double *>(base + 256);
  int *coords = reinterpret_cast<int *>(base + 256 + cbytes);
  double *values = reinterpret_cast<double *>(base + 256 + cbytes + lbytes);
  unsigned host_counters[2] = {0u, 0u};
  auto run = [&]() -> int {
    hipStream_t s = c.stream;
    PSH_HIP(hipMemsetAsync(counters, 0, 2 * sizeof(unsigned), s));
    auto blocks = [](size_t lines) { return dim3(static_cast<unsigned>((lines + kBlobThreads - 1) / kBlobThreads)); };
    // maximum_filter1d along the image rows' index (axis 0 of scikit-image's (m, n, K) cube), the columns' (axis 1), the scales (axis 2)
    hipLaunchKernelGGL(blob_max3_lines, blocks(static_cast<size_t>(K) * n), dim3(kBlobThreads), 0, s, cube_dev, cmax,
                       static_cast<size_t>(K) * n, static_cast<size_t>(n), plane, m, static_cast<size_t>(n));
    hipLaunchKernelGGL(blob_max3_lines, blocks(static_cast<size_t>(K) * m), dim3(kBlobThreads), 0, s, static_cast<const double *>(cmax), cmax,
                       static_cast<size_t>(K) * m, static_cast<size_t>(1), static_cast<size_t>(n), n, static_cast<size_t>(1));
    hipLaunchKernelGGL(blob_max3_lines, blocks(plane), dim3(kBlobThreads), 0, s, static_cast<const double *>(cmax), cmax, plane,
                       plane, static_cast<size_t>(0), K, plane);
    const unsigned grid = static_cast<unsigned>(std::min<size_t>((total + kBlobThreads - 1) / kBlobThreads, 8192));
    hipLaunchKernelGGL(blob_collect, dim3(grid), dim3(kBlobThreads), 0, s, cube_dev, static_cast<const double *>(cmax), K, m, n, threshold,
                       capacity, coords, values, counters);
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(host_counters, counters, sizeof(host_counters), hipMemcpyDeviceToHost, s));
    PSH_HIP(hipStreamSynchronize(s));
    if (host_counters[1] == 0u) {  // peak.py _get_peak_mask: "no peak for a trivial image"
      *count_host = 0;
      return PSH_OK;
    }
    *count_host = static_cast<int>(host_counters[0]);
    const unsigned got = std::min<unsigned>(host_counters[0], static_cast<unsigned>(capacity));
    if (got) {
      PSH_HIP(hipMemcpyAsync(coords_host, coords, static_cast<size_t>(got) * 3 * sizeof(int), hipMemcpyDeviceToHost, s));
      PSH_HIP(hipMemcpyAsync(values_host, values, static_cast<size_t>(got) * sizeof(double), hipMemcpyDeviceToHost, s));
      PSH_HIP(hipStreamSynchronize(s));
    }
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  return rc;
}

extern "C" int psh_blob_gather_dev(const double *plane_dev, int m, int n, const int *yx_host, int count, double *values_host) {
  using namespace psh;
  PSH_REQUIRE_INIT();
  if (!plane_dev || !yx_host || !values_host) return fail(PSH_EINVAL, "blob_gather: NULL pointer");
  if (count <= 0) return PSH_OK;
  for (int i = 0; i < count; ++i)
    if (yx_host[2 * i] < 0 || yx_host[2 * i] >= m || yx_host[2 * i + 1] < 0 || yx_host[2 * i + 1] >= n)
      return fail(PSH_EINVAL, "blob_gather: pixel (%d,%d) outside the (%d,%d) plane", yx_host[2 * i], yx_host[2 * i + 1], m, n);
  Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t ibytes = (static_cast<size_t>(count) * 2 * sizeof(int) + 255) & ~static_cast<size_t>(255);
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, ibytes + static_cast<size_t>(count) * sizeof(double))) return rc;
  int *yx = static_cast<int *>(blk);
  double *vals = reinterpret_cast<double *>(static_cast<char *>(blk) + ibytes);
  auto run = [&]() -> int {
    PSH_HIP(hipMemcpyAsync(yx, yx_host, static_cast<size_t>(count) * 2 * sizeof(int), hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(blob_gather, dim3((count + 255) / 256), dim3(256), 0, c.stream, plane_dev, n, static_cast<const int *>(yx), count, vals);
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(values_host, vals, static_cast<size_t>(count) * sizeof(double), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  return rc;
}
