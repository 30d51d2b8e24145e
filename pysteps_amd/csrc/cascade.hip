// Spectral building blocks of the STEPS member loop on the device.
//
//  * psh_cascade_decompose_dev - pysteps/cascade/decomposition.py:77-262 (decomposition_fft) for the
//    form the STEPS loop uses (spatial in, spatial out, no mask): rfft2 once, then per cascade level
//    irfft2(spectrum x band-pass weights) with the weights applied inside the column pass of the
//    inverse transform, mean / standard deviation of every level (np.mean, np.std: population
//    standard deviation) and the optional normalisation.
//  * psh_cascade_recompose_dev - decomposition.py:265-305 (recompose_fft): sum_k level_k * sigma_k +
//    mu_k (+ field mean).
//  * psh_noise_filter_dev      - pysteps/noise/fftgenerators.py:412-433, the transform part of
//    generate_noise_2d_fft_filter: irfft2(rfft2(white noise) x filter), standardised to zero mean /
//    unit variance.  The white noise itself comes from the caller's numpy RandomState (the
//    reference's random stream is part of its result).
//  * psh_ar_iterate_dev        - pysteps/timeseries/autoregression.py:1020-1070 (iterate_ar_model): one step of
//    the AR(p) model of a cascade level, bit-identical with the NumPy expression.
// Everything is float64 like the reference; the reductions accumulate per block and are finished
// by one block in a fixed order (deterministic, no atomics on values).
#include "common.h"

extern "C" int psh_fft_rfft2_dev(const double *in_dev, int m, int n, void *out_dev);

namespace psh {
namespace {

constexpr int kRedBlocksF64 = 512;
constexpr int kRedThreads = 256;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// per plane p (gridDim.y): partial[p][block] = {sum, sum of squares} of (x - shift[p]); shift is a
// device value per plane (nullptr: 0) that keeps the squares small (the level's first element)
__global__ __launch_bounds__(kRedThreads) void moments_partial(const double *__restrict__ x, size_t plane,
                                                               double2 *__restrict__ partial) {
  __shared__ double2 s_part[kRedThreads / 64];
  const double *src = x + static_cast<size_t>(blockIdx.y) * plane;
  const double shift = src[0];
  double s = 0.0, q = 0.0;
  const size_t stride = static_cast<size_t>(gridDim.x) * kRedThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kRedThreads + threadIdx.x; i < plane; i += stride) {
    const double v = src[i] - shift;
    s += v;
    q += v * v;
  }
  s = wave_sum_f64(s);
  q = wave_sum_f64(q);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = make_double2(s, q);
  __syncthreads();
  if (threadIdx.x == 0) {
    double2 t = s_part[0];
    for (int w = 1; w < kRedThreads / 64; ++w) {
      t.x += s_part[w].x;
      t.y += s_part[w].y;
    }
    partial[static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x] = t;
  }
}

// stats[p] = {mean, std} (population), one block per plane
__global__ __launch_bounds__(kRedThreads) void moments_final(const double2 *__restrict__ partial, int nblocks,
                                                             const double *__restrict__ x, size_t plane,
                                                             double2 *__restrict__ stats) {
  __shared__ double2 s_part[kRedThreads / 64];
  const double2 *src = partial + static_cast<size_t>(blockIdx.x) * nblocks;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += kRedThreads) {
    s += src[i].x;
    q += src[i].y;
  }
  s = wave_sum_f64(s);
  q = wave_sum_f64(q);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = make_double2(s, q);
  __syncthreads();
  if (threadIdx.x == 0) {
    double2 t = s_part[0];
    for (int w = 1; w < kRedThreads / 64; ++w) {
      t.x += s_part[w].x;
      t.y += s_part[w].y;
    }
    const double shift = x[static_cast<size_t>(blockIdx.x) * plane];
    const double cnt = static_cast<double>(plane);
    const double mean_shifted = t.x / cnt;
    const double var = fmax(t.y / cnt - mean_shifted * mean_shifted, 0.0);
    stats[blockIdx.x] = make_double2(mean_shifted + shift, sqrt(var));
  }
}

// x[p] = (x[p] - mean[p]) / std[p], or x - mean only (centre_only)
__global__ __launch_bounds__(kRedThreads) void standardise(double *__restrict__ x, size_t plane,
                                                           const double2 *__restrict__ stats, int centre_only) {
  double *dst = x + static_cast<size_t>(blockIdx.y) * plane;
  const double2 st = stats[blockIdx.y];
  const size_t stride = static_cast<size_t>(gridDim.x) * kRedThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kRedThreads + threadIdx.x; i < plane; i += stride)
    dst[i] = centre_only ? dst[i] - st.x : (dst[i] - st.x) / st.y;
}

// decomposition.py:294-301: sum_k (levels[k] * sigma[k] + mu[k]) (+ field mean), in NumPy's order and
// rounding - the product and the sum of every level are separate array operations, np.sum over the
// stacked levels adds them one after the other (the reduction axis is the outer one: no pairwise
// blocks), the field mean comes last - so the result is bit-identical with the reference's
__global__ __launch_bounds__(kRedThreads) void recompose(const double *__restrict__ levels, int nlevels,
                                                         size_t plane, const double *__restrict__ musigma,
                                                         double add, int has_add, double *__restrict__ out) {
#pragma clang fp contract(off)
  const size_t stride = static_cast<size_t>(gridDim.x) * kRedThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kRedThreads + threadIdx.x; i < plane; i += stride) {
    double acc = 0.0;
    for (int k = 0; k < nlevels; ++k) {
      double v = levels[static_cast<size_t>(k) * plane + i];
      if (musigma) {
        const double scaled = v * musigma[2 * k + 1];
        v = scaled + musigma[2 * k];
      }
      acc = k == 0 ? v : acc + v;
    }
    out[i] = has_add ? acc + add : acc;
  }
}

// pysteps/timeseries/autoregression.py:1056-1070: x_new = 0.0 + phi_1 x[-1] + phi_2 x[-2] + ... (+ phi_{p+1} eps),
// every product rounded before it is added (NumPy evaluates `x_new += phi[i] * x[-(i + 1)]` as two array
// operations), result = the series moved up by one with x_new at its end
constexpr int kArMaxOrder = 8;
struct ArPhi {
  double phi[kArMaxOrder + 1];
  int p, has_eps;
};

__global__ __launch_bounds__(kRedThreads) void ar_iterate(const double *__restrict__ x, int nt, size_t plane, ArPhi a,
                                                          const double *__restrict__ eps, double *__restrict__ out) {
#pragma clang fp contract(off)
  const size_t stride = static_cast<size_t>(gridDim.x) * kRedThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kRedThreads + threadIdx.x; i < plane; i += stride) {
    double acc = 0.0;
    for (int j = 0; j < a.p; ++j) {
      const double term = a.phi[j] * x[static_cast<size_t>(nt - 1 - j) * plane + i];
      acc = acc + term;
    }
    if (a.has_eps) {
      const double term = a.phi[a.p] * eps[i];
      acc = acc + term;
    }
    for (int k = 1; k < nt; ++k) out[static_cast<size_t>(k - 1) * plane + i] = x[static_cast<size_t>(k) * plane + i];
    out[static_cast<size_t>(nt - 1) * plane + i] = acc;
  }
}

int moments(const double *x_dev, int planes, size_t plane, double2 *stats_dev, hipStream_t stream) {
  void *partial = nullptr;
  if (int rc = psh_malloc(&partial, static_cast<size_t>(planes) * kRedBlocksF64 * sizeof(double2))) return rc;
  hipLaunchKernelGGL(moments_partial, dim3(kRedBlocksF64, planes), dim3(kRedThreads), 0, stream, x_dev, plane,
                     static_cast<double2 *>(partial));
  hipLaunchKernelGGL(moments_final, dim3(planes), dim3(kRedThreads), 0, stream, static_cast<const double2 *>(partial),
                     kRedBlocksF64, x_dev, plane, stats_dev);
  const hipError_t e = hipGetLastError();
  (void)psh_free(partial);
  if (e != hipSuccess) return fail(PSH_EHIP, "moments launch failed: %s", hipGetErrorString(e));
  return PSH_OK;
}

}  // namespace
}  // namespace psh

using psh::fail;

// stats_out_dev (nullable): the level statistics (mean, std per level: nlevels double2) stay on the device
static int cascade_decompose_run(const double *field_dev, const double *weights_dev, int nlevels, int m, int n,
                                 int normalize, int subtract_mean, double *levels_dev, double *means_host,
                                 double *stds_host, double *field_mean_host, double *stats_out_dev) {
  PSH_REQUIRE_INIT();
  if (!field_dev || !weights_dev || !levels_dev) return fail(PSH_EINVAL, "cascade_decompose: NULL pointer");
  if ((means_host == nullptr) != (stds_host == nullptr))
    return fail(PSH_EINVAL, "cascade_decompose: means and stds go together");
  if (!means_host && subtract_mean && field_mean_host)
    return fail(PSH_EINVAL, "cascade_decompose: the field mean is returned with the level statistics only");
  if (nlevels < 1 || nlevels > 64) return fail(PSH_EINVAL, "cascade_decompose: 1..64 cascade levels");
  if (!psh::fft_shape_supported(m, n))
    return fail(PSH_EUNSUPPORTED, "cascade_decompose: (%d,%d) - sides: powers of two up to 8192 or any length up to 4096", m, n);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * n, nc = static_cast<size_t>(n / 2 + 1);
  const size_t spec_bytes = static_cast<size_t>(m) * nc * sizeof(double2);
  void *blk = nullptr;
  // [spectrum | scratch of the inverse column pass | centred copy of the field (subtract_mean) | stats]
  const size_t stats_bytes = (static_cast<size_t>(nlevels) + 1) * sizeof(double2);
  if (int rc = psh_malloc(&blk, 2 * spec_bytes + (subtract_mean ? plane * sizeof(double) : 0) + stats_bytes)) return rc;
  char *base = static_cast<char *>(blk);
  void *spec = base, *scratch = base + spec_bytes;
  double *centred = reinterpret_cast<double *>(base + 2 * spec_bytes);
  double2 *stats = reinterpret_cast<double2 *>(base + 2 * spec_bytes + (subtract_mean ? plane * sizeof(double) : 0));
  auto run = [&]() -> int {
    const double *src = field_dev;
    if (subtract_mean) {  // decomposition.py:199-202
      PSH_HIP(hipMemcpyAsync(centred, field_dev, plane * sizeof(double), hipMemcpyDeviceToDevice, c.stream));
      if (int rc = psh::moments(centred, 1, plane, stats + nlevels, c.stream)) return rc;
      hipLaunchKernelGGL(psh::standardise, dim3(psh::kRedBlocksF64, 1), dim3(psh::kRedThreads), 0, c.stream, centred,
                         plane, stats + nlevels, 1);
      src = centred;
    }
    if (int rc = psh_fft_rfft2_dev(src, m, n, spec)) return rc;
    for (int k = 0; k < nlevels; ++k) {  // :210-215: field_fft * weights_2d[k], back to the spatial domain
      if (int rc = psh::fft_irfft2_weighted(spec, weights_dev + static_cast<size_t>(k) * m * nc, m, n,
                                            levels_dev + static_cast<size_t>(k) * plane, scratch))
        return rc;
    }
    if (int rc = psh::moments(levels_dev, nlevels, plane, stats, c.stream)) return rc;  // :217-232
    if (stats_out_dev)
      PSH_HIP(hipMemcpyAsync(stats_out_dev, stats, static_cast<size_t>(nlevels) * sizeof(double2), hipMemcpyDeviceToDevice, c.stream));
    if (normalize)
      hipLaunchKernelGGL(psh::standardise, dim3(psh::kRedBlocksF64, nlevels), dim3(psh::kRedThreads), 0, c.stream,
                         levels_dev, plane, stats, 0);
    PSH_HIP(hipGetLastError());
    if (!means_host) return PSH_OK;  // resident member loop: nobody on the host needs the statistics, no wait
    double2 host[65];
    PSH_HIP(hipMemcpyAsync(host, stats, stats_bytes, hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    for (int k = 0; k < nlevels; ++k) {
      means_host[k] = host[k].x;
      stds_host[k] = host[k].y;
    }
    if (field_mean_host) *field_mean_host = subtract_mean ? host[nlevels].x : 0.0;
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  return rc;
}

extern "C" int psh_cascade_decompose_dev(const double *field_dev, const double *weights_dev, int nlevels, int m,
                                         int n, int normalize, int subtract_mean, double *levels_dev,
                                         double *means_host, double *stds_host, double *field_mean_host) {
  return cascade_decompose_run(field_dev, weights_dev, nlevels, m, n, normalize, subtract_mean, levels_dev, means_host,
                               stds_host, field_mean_host, nullptr);
}

// the levels as the transforms leave them + their statistics in device memory, nothing waits: the
// consumer (psh_steps_ar_recompose_raw_dev) standardises on the way in
extern "C" int psh_cascade_decompose_stats_dev(const double *field_dev, const double *weights_dev, int nlevels, int m,
                                               int n, double *levels_dev, double *stats_dev) {
  if (!stats_dev) return fail(PSH_EINVAL, "cascade_decompose_stats: NULL pointer");
  return cascade_decompose_run(field_dev, weights_dev, nlevels, m, n, 0, 0, levels_dev, nullptr, nullptr, nullptr, stats_dev);
}

extern "C" int psh_cascade_recompose_dev(const double *levels_dev, int nlevels, int m, int n, const double *means_host,
                                         const double *stds_host, double field_mean, double *out_dev) {
  PSH_REQUIRE_INIT();
  if (!levels_dev || !out_dev) return fail(PSH_EINVAL, "cascade_recompose: NULL pointer");
  if (nlevels < 1 || nlevels > 64 || m <= 0 || n <= 0) return fail(PSH_EINVAL, "cascade_recompose: invalid shape");
  if ((means_host == nullptr) != (stds_host == nullptr))
    return fail(PSH_EINVAL, "cascade_recompose: means and stds go together");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const double *musigma = nullptr;
  if (means_host) {  // per-call constants through the pinned slot ring (floats: two per double)
    float *h = nullptr;
    const float *d = nullptr;
    if (int rc = psh::const_slot(&h, &d)) return rc;
    double *hd = reinterpret_cast<double *>(h);
    for (int k = 0; k < nlevels; ++k) {
      hd[2 * k] = means_host[k];
      hd[2 * k + 1] = stds_host[k];
    }
    PSH_HIP(hipMemcpyAsync(const_cast<float *>(d), h, static_cast<size_t>(nlevels) * 2 * sizeof(double),
                           hipMemcpyHostToDevice, c.stream));
    musigma = reinterpret_cast<const double *>(d);
  }
  hipLaunchKernelGGL(psh::recompose, dim3(2048), dim3(psh::kRedThreads), 0, c.stream, levels_dev, nlevels,
                     static_cast<size_t>(m) * n, musigma, field_mean, field_mean != 0.0 ? 1 : 0, out_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_noise_filter_dev(const double *white_dev, const double *filter_dev, int m, int n, double *out_dev) {
  PSH_REQUIRE_INIT();
  if (!white_dev || !filter_dev || !out_dev) return fail(PSH_EINVAL, "noise_filter: NULL pointer");
  if (!psh::fft_shape_supported(m, n))
    return fail(PSH_EUNSUPPORTED, "noise_filter: (%d,%d) - sides: powers of two up to 8192 or any length up to 4096", m, n);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * n;
  const size_t spec_bytes = static_cast<size_t>(m) * (n / 2 + 1) * sizeof(double2);
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, 2 * spec_bytes + sizeof(double2))) return rc;
  char *base = static_cast<char *>(blk);
  double2 *stats = reinterpret_cast<double2 *>(base + 2 * spec_bytes);
  auto run = [&]() -> int {
    if (int rc = psh_fft_rfft2_dev(white_dev, m, n, base)) return rc;                                  // :423
    if (int rc = psh::fft_irfft2_weighted(base, filter_dev, m, n, out_dev, base + spec_bytes)) return rc;  // :427-431
    if (int rc = psh::moments(out_dev, 1, plane, stats, c.stream)) return rc;
    hipLaunchKernelGGL(psh::standardise, dim3(psh::kRedBlocksF64, 1), dim3(psh::kRedThreads), 0, c.stream, out_dev,
                       plane, stats, 0);                                                                // :432
    PSH_HIP(hipGetLastError());
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);  // stream-ordered
  return rc;
}

extern "C" int psh_ar_iterate_dev(const double *x_dev, int nt, size_t plane, const double *phi_host, int p,
                                  const double *eps_dev, double *out_dev) {
  PSH_REQUIRE_INIT();
  if (!x_dev || !phi_host || !out_dev) return fail(PSH_EINVAL, "ar_iterate: NULL pointer");
  if (plane == 0 || nt < 1) return fail(PSH_EINVAL, "ar_iterate: empty series");
  if (p < 1 || p > psh::kArMaxOrder) return fail(PSH_EUNSUPPORTED, "ar_iterate: AR order 1..%d", psh::kArMaxOrder);
  if (nt < p)  // autoregression.py:1041-1045
    return fail(PSH_EINVAL, "dimension mismatch between x and phi: x.shape[0]=%d, len(phi)=%d", nt, p + 1);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::ArPhi a;
  for (int j = 0; j <= p; ++j) a.phi[j] = phi_host[j];
  a.p = p;
  a.has_eps = eps_dev != nullptr;
  const size_t blocks = (plane + psh::kRedThreads - 1) / psh::kRedThreads;
  hipLaunchKernelGGL(psh::ar_iterate, dim3(static_cast<unsigned>(blocks < 4096 ? blocks : 4096)),
                     dim3(psh::kRedThreads), 0, c.stream, x_dev, nt, plane, a, eps_dev, out_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}
