// Element-wise passes either side of the advection path, kept on the device so that a
// nowcast chain (rain rate -> dB -> LK -> extrapolation -> rain rate) never leaves HBM:
//   dB transform  pysteps/utils/transformation.py:150-232 (dB_transform, forward and inverse)
//   field statistics (min / max over finite values, count of non-finite values) that the
//   reference gets from NumPy scans: nowcasts/extrapolation.py:76 (allow_nonfinite_values),
//   semilagrangian.py:171-172 (outval="min").
// HBM-streaming, dwordx4 where the size allows; NaN stays NaN like in NumPy.
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

__global__ __launch_bounds__(256) void field_stats(const float *__restrict__ in, size_t n,
                                                   float *__restrict__ partial) {
  float mn = INFINITY, mx = -INFINITY, bad = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = in[i];
    if (isfinite(v)) {
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    } else {
      bad += 1.f;
    }
  }
  __shared__ float s[3][4];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, d));
    mx = fmaxf(mx, __shfl_xor(mx, d));
    bad += __shfl_xor(bad, d);
  }
  if ((threadIdx.x & 63) == 0) {
    s[0][threadIdx.x >> 6] = mn;
    s[1][threadIdx.x >> 6] = mx;
    s[2][threadIdx.x >> 6] = bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[3 * blockIdx.x] = fminf(fminf(s[0][0], s[0][1]), fminf(s[0][2], s[0][3]));
    partial[3 * blockIdx.x + 1] = fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3]));
    partial[3 * blockIdx.x + 2] = s[2][0] + s[2][1] + s[2][2] + s[2][3];
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

extern "C" int psh_field_stats_dev(const float *in_dev, size_t n, double *min_out, double *max_out,
                                   double *nonfinite_out) {
  PSH_REQUIRE_INIT();
  if (!in_dev && n) return psh::fail(PSH_EINVAL, "field_stats: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  constexpr int kBlocks = 1024;
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, 3 * kBlocks * sizeof(float))) return rc;
  float h[3 * kBlocks];
  auto run = [&]() -> int {
    hipLaunchKernelGGL(psh::field_stats, dim3(kBlocks), dim3(256), 0, c.stream, in_dev, n,
                       static_cast<float *>(blk));
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(h, blk, sizeof(h), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    return PSH_OK;
  };
  const int rc = run();
  (void)psh_free(blk);
  if (rc) return rc;
  double mn = INFINITY, mx = -INFINITY, bad = 0.0;
  for (int b = 0; b < kBlocks; ++b) {
    mn = fmin(mn, h[3 * b]);
    mx = fmax(mx, h[3 * b + 1]);
    bad += h[3 * b + 2];
  }
  if (min_out) *min_out = mn;   // +inf if there is no finite value
  if (max_out) *max_out = mx;   // -inf if there is no finite value
  if (nonfinite_out) *nonfinite_out = bad;
  return PSH_OK;
}
