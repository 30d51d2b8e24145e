// Cubic B-spline prefilter for interp_order=3 of the semi-Lagrangian extrapolator (gfx950).
//
// pysteps/extrapolation/semilagrangian.py:225-232 calls scipy.ndimage.map_coordinates(order=3,
// prefilter=True, mode="constant"); SciPy first turns the samples into B-spline coefficients
// with spline_filter (ni_splines.c): per axis  c = 6 s,  causal  c[i] += z c[i-1],  anticausal
// c[i] = z (c[i+1] - c[i]),  z = sqrt(3) - 2,  with MIRROR boundary initialisation for mode
// "constant" (_init_causal_mirror / _init_anticausal_mirror).  Missing values are zeroed first
// (semilagrangian.py:151-153).
//
// A first-order recursion is sequential along its axis, but |z| = 0.268 forgets its past in ~40
// samples (|z|^40 = 1e-23), so every column is cut into segments that start their recursion 40
// samples early: 8-16x more parallelism at ~10 % redundant work, results identical to fp32
// rounding.  Rows are filtered as columns of the transposed image (LDS-tiled transposes), so
// every memory access of the recursions is coalesced.
#include "common.h"

#pragma clang fp contract(off)

namespace psh {
namespace {

constexpr int kSeg = 512;   // samples owned by one thread
constexpr int kWarm = 40;   // recursion warm-up before the owned segment
constexpr float kPole = -0.2679491924311227f;  // sqrt(3) - 2

__global__ __launch_bounds__(256) void spline_zero_nonfinite(const float *__restrict__ in,
                                                             float *__restrict__ out, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = in[i];
    out[i] = isfinite(v) ? v : 0.f;
  }
}

// causal pass along axis 0 of a (len, width) row-major array: dst[i] = 6 src[i] + z dst[i-1]
__global__ __launch_bounds__(64) void spline_causal(const float *__restrict__ src,
                                                    float *__restrict__ dst, int len, int width) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= width) return;
  const int s0 = blockIdx.y * kSeg, s1 = min(len, s0 + kSeg);
  const float z = kPole;
  float c;
  int i;
  if (s0 <= kWarm) {
    // exact mirror initialisation at the array start (ni_splines.c _init_causal_mirror)
    if (len == 1) {
      dst[x] = src[x];  // a single sample is its own coefficient
      return;
    }
    const float zn1 = powf(fabsf(z), static_cast<float>(len - 1)) * (((len - 1) & 1) ? -1.f : 1.f);
    float acc = 6.f * src[x] + zn1 * 6.f * src[static_cast<size_t>(len - 1) * width + x];
    float zi = z;
    const int horizon = min(len - 2, 64);  // |z|^64 ~ 1e-37
    for (int k = 1; k <= horizon; ++k) {
      float term = 6.f * src[static_cast<size_t>(k) * width + x];
      if (zn1 != 0.f) term += zn1 * 6.f * src[static_cast<size_t>(len - 1 - k) * width + x];
      acc += zi * term;
      zi *= z;
    }
    c = acc / (1.f - zn1 * zn1);
    if (s0 == 0) dst[x] = c;
    i = 1;
  } else {
    i = s0 - kWarm;
    c = 6.f * src[static_cast<size_t>(i) * width + x];  // any start value: forgotten after kWarm steps
    ++i;
  }
  for (; i < s1; ++i) {
    c = 6.f * src[static_cast<size_t>(i) * width + x] + z * c;
    if (i >= s0) dst[static_cast<size_t>(i) * width + x] = c;
  }
}

// anticausal pass: dst[i] = z (dst[i+1] - cp[i]) with cp the causal result
__global__ __launch_bounds__(64) void spline_anticausal(const float *__restrict__ cp,
                                                        float *__restrict__ dst, int len,
                                                        int width) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= width || len == 1) {
    if (x < width && blockIdx.y == 0) dst[x] = cp[x];
    return;
  }
  const int s0 = blockIdx.y * kSeg, s1 = min(len, s0 + kSeg);
  const float z = kPole;
  float c;
  int i;
  if (s1 + kWarm >= len) {
    // exact mirror initialisation at the array end (_init_anticausal_mirror)
    const float last = cp[static_cast<size_t>(len - 1) * width + x];
    const float prev = cp[static_cast<size_t>(len - 2) * width + x];
    c = (z * prev + last) * z / (z * z - 1.f);
    if (s1 == len) dst[static_cast<size_t>(len - 1) * width + x] = c;
    i = len - 2;
  } else {
    i = s1 + kWarm;
    c = 0.f;  // forgotten after kWarm steps
  }
  for (; i >= s0; --i) {
    c = z * (c - cp[static_cast<size_t>(i) * width + x]);
    if (i < s1) dst[static_cast<size_t>(i) * width + x] = c;
  }
}

__global__ __launch_bounds__(256) void transpose32(const float *__restrict__ in, float *__restrict__ out,
                                                   int rows, int cols) {
  __shared__ float tile[32][33];
  const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int x = x0 + tx, y = y0 + j;
    if (x < cols && y < rows) tile[j][tx] = in[static_cast<size_t>(y) * cols + x];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int x = y0 + tx, y = x0 + j;  // transposed coordinates
    if (x < rows && y < cols) out[static_cast<size_t>(y) * rows + x] = tile[tx][j];
  }
}

}  // namespace

// coef <- cubic B-spline coefficients of precip (NaN/Inf -> 0); tmp: second (m,n) plane.
hipError_t spline_prefilter(const float *precip, float *coef, float *tmp, int m, int n,
                            hipStream_t stream) {
  const size_t npx = static_cast<size_t>(m) * n;
  hipLaunchKernelGGL(spline_zero_nonfinite, dim3(2048), dim3(256), 0, stream, precip, coef, npx);
  // axis 0 (columns of the image)
  dim3 g0((n + 63) / 64, (m + kSeg - 1) / kSeg);
  hipLaunchKernelGGL(spline_causal, g0, dim3(64), 0, stream, coef, tmp, m, n);
  hipLaunchKernelGGL(spline_anticausal, g0, dim3(64), 0, stream, tmp, coef, m, n);
  // axis 1 (rows) as columns of the transpose
  hipLaunchKernelGGL(transpose32, dim3((n + 31) / 32, (m + 31) / 32), dim3(256), 0, stream, coef, tmp, m, n);
  dim3 g1((m + 63) / 64, (n + kSeg - 1) / kSeg);
  hipLaunchKernelGGL(spline_causal, g1, dim3(64), 0, stream, tmp, coef, n, m);
  hipLaunchKernelGGL(spline_anticausal, g1, dim3(64), 0, stream, coef, tmp, n, m);
  hipLaunchKernelGGL(transpose32, dim3((m + 31) / 32, (n + 31) / 32), dim3(256), 0, stream, tmp, coef, n, m);
  return hipGetLastError();
}

}  // namespace psh
