// Element-wise passes either side of the advection path, kept on the device so that a
// nowcast chain (rain rate -> dB -> LK -> extrapolation -> rain rate) never leaves HBM:
//   dB transform  pysteps/utils/transformation.py:150-232 (dB_transform, forward and inverse)
//   field statistics (min / max over finite values, count of non-finite values) that the
//   reference gets from NumPy scans: nowcasts/extrapolation.py:76 (allow_nonfinite_values),
//   semilagrangian.py:171-172 (outval="min").
// HBM-streaming, dwordx4 where the size allows; NaN stays NaN like in NumPy.
#include <algorithm>
#include <cmath>

#include "common.h"

namespace psh {
namespace {

__device__ __forceinline__ float to_db(float r, float thr, float zerovalue) {
  // R[~(R < thr)] = 10 log10(R); R[R < thr] = zerovalue  (NaN compares false -> log10(NaN) = NaN)
  return r < thr ? zerovalue : 10.0f * log10f(r);
}
__device__ __forceinline__ float from_db(float r, float thr_lin, float zerovalue) {
  const float v = exp10f(r / 10.0f);
  return v < thr_lin ? zerovalue : v;
}

template <bool INVERSE>
__global__ __launch_bounds__(256) void db_transform(const float *__restrict__ in,
                                                    float *__restrict__ out, size_t n, float thr,
                                                    float zerovalue) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  const size_t n4 = n / 4;
  const float4 *in4 = reinterpret_cast<const float4 *>(in);
  float4 *out4 = reinterpret_cast<float4 *>(out);
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = in4[i];
    v.x = INVERSE ? from_db(v.x, thr, zerovalue) : to_db(v.x, thr, zerovalue);
    v.y = INVERSE ? from_db(v.y, thr, zerovalue) : to_db(v.y, thr, zerovalue);
    v.z = INVERSE ? from_db(v.z, thr, zerovalue) : to_db(v.z, thr, zerovalue);
    v.w = INVERSE ? from_db(v.w, thr, zerovalue) : to_db(v.w, thr, zerovalue);
    out4[i] = v;
  }
  for (size_t i = n4 * 4 + static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = INVERSE ? from_db(in[i], thr, zerovalue) : to_db(in[i], thr, zerovalue);
}

constexpr int kStatFields = 5;

// float64 twin of field_stats (double partials): the host path checks float64 inputs BEFORE they are
// narrowed to float32 - a finite value beyond the float32 range must not read as "non-finite input"
__global__ __launch_bounds__(256) void field_stats_f64(const double *__restrict__ in, size_t n,
                                                       double *__restrict__ partial) {
  double mn = INFINITY, mx = -INFINITY, bad = 0.0, ninf = 0.0, pinf = 0.0;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double v = in[i];
    if (isfinite(v)) {
      mn = fmin(mn, v);
      mx = fmax(mx, v);
    } else {
      bad += 1.0;
      ninf += v == -INFINITY ? 1.0 : 0.0;
      pinf += v == INFINITY ? 1.0 : 0.0;
    }
  }
  __shared__ double s[kStatFields][4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mn = fmin(mn, __shfl_xor(mn, d));
    mx = fmax(mx, __shfl_xor(mx, d));
    bad += __shfl_xor(bad, d);
    ninf += __shfl_xor(ninf, d);
    pinf += __shfl_xor(pinf, d);
  }
  if ((threadIdx.x & 63) == 0) {
    s[0][threadIdx.x >> 6] = mn;
    s[1][threadIdx.x >> 6] = mx;
    s[2][threadIdx.x >> 6] = bad;
    s[3][threadIdx.x >> 6] = ninf;
    s[4][threadIdx.x >> 6] = pinf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double *o = partial + kStatFields * blockIdx.x;
    o[0] = fmin(fmin(s[0][0], s[0][1]), fmin(s[0][2], s[0][3]));
    o[1] = fmax(fmax(s[1][0], s[1][1]), fmax(s[1][2], s[1][3]));
    for (int k = 2; k < kStatFields; ++k) o[k] = s[k][0] + s[k][1] + s[k][2] + s[k][3];
  }
}

__global__ __launch_bounds__(256) void field_stats(const float *__restrict__ in, size_t n,
                                                   float *__restrict__ partial) {
  // per block: min / max over finite values, counts of non-finite values, of -inf and of +inf
  // (counts < 2^24 per block: exact in float)
  float mn = INFINITY, mx = -INFINITY, bad = 0.f, ninf = 0.f, pinf = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = in[i];
    if (isfinite(v)) {
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    } else {
      bad += 1.f;
      ninf += v == -INFINITY ? 1.f : 0.f;
      pinf += v == INFINITY ? 1.f : 0.f;
    }
  }
  __shared__ float s[kStatFields][4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, d));
    mx = fmaxf(mx, __shfl_xor(mx, d));
    bad += __shfl_xor(bad, d);
    ninf += __shfl_xor(ninf, d);
    pinf += __shfl_xor(pinf, d);
  }
  if ((threadIdx.x & 63) == 0) {
    s[0][threadIdx.x >> 6] = mn;
    s[1][threadIdx.x >> 6] = mx;
    s[2][threadIdx.x >> 6] = bad;
    s[3][threadIdx.x >> 6] = ninf;
    s[4][threadIdx.x >> 6] = pinf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float *o = partial + kStatFields * blockIdx.x;
    o[0] = fminf(fminf(s[0][0], s[0][1]), fminf(s[0][2], s[0][3]));
    o[1] = fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3]));
    for (int k = 2; k < kStatFields; ++k) o[k] = s[k][0] + s[k][1] + s[k][2] + s[k][3];
  }
}

// number of elements > thr (NaN compares false, like NumPy): the rain-pixel count of
// pysteps/utils/check_norain.py:48-49
// the comparison runs in double: the caller decides what the threshold is rounded to first (NumPy
// compares a float32 array with a Python float in float32, with a numpy.float64 scalar in float64)
__global__ __launch_bounds__(256) void count_above(const float *__restrict__ in, size_t n, double thr,
                                                   unsigned long long *__restrict__ total) {
  unsigned cnt = 0;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    cnt += static_cast<double>(in[i]) > thr ? 1u : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(total, static_cast<unsigned long long>(cnt));  // integer: order-free
}

template <typename Tin, typename Tout>
__global__ __launch_bounds__(256) void convert_elements(const Tin *__restrict__ in, Tout *__restrict__ out, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x * 4;
  for (size_t i = (static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      Tin v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = in[i + j];
#pragma unroll
      for (int j = 0; j < 4; ++j) out[i + j] = static_cast<Tout>(v[j]);
    } else {
      for (size_t k = i; k < n; ++k) out[k] = static_cast<Tout>(in[k]);
    }
  }
}

}  // namespace
}  // namespace psh

extern "C" int psh_db_transform_dev(const float *in_dev, float *out_dev, size_t n, double threshold,
                                    double zerovalue, int inverse) {
  PSH_REQUIRE_INIT();
  if (n == 0) return PSH_OK;
  if (!in_dev || !out_dev) return psh::fail(PSH_EINVAL, "db_transform: NULL pointer");
  if ((reinterpret_cast<uintptr_t>(in_dev) | reinterpret_cast<uintptr_t>(out_dev)) & 15)
    return psh::fail(PSH_EINVAL, "db_transform: buffers must be 16-byte aligned");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const dim3 grid(c.cu_count * 8), block(256);
  if (inverse) {
    // threshold is given in dB and compared in linear units (transformation.py:225-226)
    const float thr_lin = static_cast<float>(pow(10.0, threshold / 10.0));
    hipLaunchKernelGGL(psh::db_transform<true>, grid, block, 0, c.stream, in_dev, out_dev, n, thr_lin,
                       static_cast<float>(zerovalue));
  } else {
    hipLaunchKernelGGL(psh::db_transform<false>, grid, block, 0, c.stream, in_dev, out_dev, n,
                       static_cast<float>(threshold), static_cast<float>(zerovalue));
  }
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

namespace psh {

hipError_t launch_convert_f64_f32(const double *in, float *out, size_t n, hipStream_t stream) {
  const unsigned grid = static_cast<unsigned>(std::min<size_t>((n / 4 + 255) / 256 + 1, 65535));
  hipLaunchKernelGGL((convert_elements<double, float>), dim3(grid), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

hipError_t launch_convert_f32_f64(const float *in, double *out, size_t n, hipStream_t stream) {
  const unsigned grid = static_cast<unsigned>(std::min<size_t>((n / 4 + 255) / 256 + 1, 65535));
  hipLaunchKernelGGL((convert_elements<float, double>), dim3(grid), dim3(256), 0, stream, in, out, n);
  return hipGetLastError();
}

// synchronous: waits for the library stream
int field_stats_full(const float *in_dev, size_t n, FieldStats *st) {
  Context &c = ctx();
  constexpr int kBlocks = 1024;
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, kStatFields * kBlocks * sizeof(float))) return rc;
  static thread_local float h[kStatFields * kBlocks];
  auto run = [&]() -> int {
    hipLaunchKernelGGL(field_stats, dim3(kBlocks), dim3(256), 0, c.stream, in_dev, n, static_cast<float *>(blk));
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(h, blk, sizeof(h), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  if (rc) return rc;
  FieldStats r;
  for (int b = 0; b < kBlocks; ++b) {
    const float *o = h + kStatFields * b;
    r.min_finite = fmin(r.min_finite, o[0]);
    r.max_finite = fmax(r.max_finite, o[1]);
    r.nonfinite += o[2];
    r.neg_inf += o[3];
    r.pos_inf += o[4];
  }
  r.count = static_cast<double>(n);
  *st = r;
  return PSH_OK;
}

int field_stats_full_f64(const double *in_dev, size_t n, FieldStats *st) {
  Context &c = ctx();
  constexpr int kBlocks = 1024;
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, kStatFields * kBlocks * sizeof(double))) return rc;
  static thread_local double h[kStatFields * kBlocks];
  auto run = [&]() -> int {
    hipLaunchKernelGGL(field_stats_f64, dim3(kBlocks), dim3(256), 0, c.stream, in_dev, n, static_cast<double *>(blk));
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(h, blk, sizeof(h), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  if (rc) return rc;
  FieldStats r;
  for (int b = 0; b < kBlocks; ++b) {
    const double *o = h + kStatFields * b;
    r.min_finite = fmin(r.min_finite, o[0]);
    r.max_finite = fmax(r.max_finite, o[1]);
    r.nonfinite += o[2];
    r.neg_inf += o[3];
    r.pos_inf += o[4];
  }
  r.count = static_cast<double>(n);
  *st = r;
  return PSH_OK;
}

}  // namespace psh

extern "C" int psh_nonfinite_count_f64_dev(const double *in_dev, size_t n, double *count_out) {
  PSH_REQUIRE_INIT();
  if ((!in_dev && n) || !count_out) return psh::fail(PSH_EINVAL, "nonfinite_count: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::FieldStats st;
  if (int rc = psh::field_stats_full_f64(in_dev, n, &st)) return rc;
  *count_out = st.nonfinite;
  return PSH_OK;
}

extern "C" int psh_count_above_dev(const float *in_dev, size_t n, double threshold, double *count_out,
                                   double *nanmin_out) {
  PSH_REQUIRE_INIT();
  if ((!in_dev && n) || !count_out) return psh::fail(PSH_EINVAL, "count_above: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::FieldStats st;
  if (int rc = psh::field_stats_full(in_dev, n, &st)) return rc;
  const double lowest = st.nanmin();
  if (nanmin_out) *nanmin_out = lowest;
  // threshold NaN = "use the minimum of the field" (precip_thr=None, check_norain.py:46-47)
  const double thr = threshold != threshold ? lowest : threshold;
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, sizeof(unsigned long long))) return rc;
  unsigned long long h = 0;
  auto run = [&]() -> int {
    PSH_HIP(hipMemsetAsync(blk, 0, sizeof(unsigned long long), c.stream));
    hipLaunchKernelGGL(psh::count_above, dim3(c.cu_count * 8), dim3(256), 0, c.stream, in_dev, n, thr,
                       static_cast<unsigned long long *>(blk));
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(&h, blk, sizeof(h), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  if (rc) return rc;
  *count_out = static_cast<double>(h);
  return PSH_OK;
}

namespace psh {
namespace {
__global__ __launch_bounds__(256) void axpy_f64(double *__restrict__ dst, const double *__restrict__ src, double alpha, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += alpha * src[i];
}
}  // namespace
}  // namespace psh

extern "C" int psh_axpy_f64_dev(double *dst_dev, const double *src_dev, double alpha, size_t n) {
  PSH_REQUIRE_INIT();
  if (n == 0) return PSH_OK;
  if (!dst_dev || !src_dev) return psh::fail(PSH_EINVAL, "axpy: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const unsigned blocks = static_cast<unsigned>(std::min<size_t>((n + 255) / 256, static_cast<size_t>(c.cu_count) * 16));
  hipLaunchKernelGGL(psh::axpy_f64, dim3(blocks), dim3(256), 0, c.stream, dst_dev, src_dev, alpha, n);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_convert_dev(const void *in_dev, void *out_dev, size_t n, int to_f64) {
  PSH_REQUIRE_INIT();
  if (n == 0) return PSH_OK;
  if (!in_dev || !out_dev) return psh::fail(PSH_EINVAL, "convert: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (to_f64)
    PSH_HIP(psh::launch_convert_f32_f64(static_cast<const float *>(in_dev), static_cast<double *>(out_dev), n, c.stream));
  else
    PSH_HIP(psh::launch_convert_f64_f32(static_cast<const double *>(in_dev), static_cast<float *>(out_dev), n, c.stream));
  return PSH_OK;
}

extern "C" int psh_field_stats_dev(const float *in_dev, size_t n, double *min_out, double *max_out,
                                   double *nonfinite_out) {
  PSH_REQUIRE_INIT();
  if (!in_dev && n) return psh::fail(PSH_EINVAL, "field_stats: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::FieldStats st;
  if (int rc = psh::field_stats_full(in_dev, n, &st)) return rc;
  if (min_out) *min_out = st.min_finite;   // +inf if there is no finite value
  if (max_out) *max_out = st.max_finite;   // -inf if there is no finite value
  if (nonfinite_out) *nonfinite_out = st.nonfinite;
  return PSH_OK;
}
