// B-spline prefilter for interp_order 2 .. 5 of the semi-Lagrangian extrapolator (gfx950); written for the cubic
// case (one pole, gain 6), the other orders pass their poles and the gain of all of them through the same kernels
// (ni_splines.c get_filter_poles / apply_filter: order 2 one pole sqrt(8) - 3; orders 4 and 5 two poles each - one
// causal / anticausal pair per pole, the gain prod (1 - z)(1 - 1/z) on the first).
//
// pysteps/extrapolation/semilagrangian.py:225-232 calls scipy.ndimage.map_coordinates(order=3,
// prefilter=True, mode="constant"); SciPy first turns the samples into B-spline coefficients
// with spline_filter (ni_splines.c): per axis  c = 6 s,  causal  c[i] += z c[i-1],  anticausal
// c[i] = z (c[i+1] - c[i]),  z = sqrt(3) - 2,  with MIRROR boundary initialisation for mode
// "constant" (_init_causal_mirror / _init_anticausal_mirror).  Missing values are zeroed first
// (semilagrangian.py:151-153).  The other map_coordinates modes (semilagrangian.py:91-96): "mirror", "wrap"
// and "grid-constant" filter with the same MIRROR boundaries, "nearest" and "reflect" with half-sample
// REFLECT boundaries (_init_*_reflect), "grid-wrap" periodically (_init_*_wrap); for "nearest" and
// "grid-constant" - no exact boundary condition in the filter - map_coordinates pads the array by 12
// samples (edge values / cval) before filtering and shifts the coordinates (_prepad_for_spline_filter).
// All of it pinned against SciPy through oracle/semilag.py.
//
// A first-order recursion is sequential along its axis, but |z| = 0.268 forgets its past in ~40
// samples (|z|^40 = 1e-23), so every column is cut into segments that start their recursion 40
// samples early: 8-16x more parallelism at ~10 % redundant work, results identical to fp32
// rounding.  Rows are filtered as columns of the transposed image (LDS-tiled transposes), so
// every memory access of the recursions is coalesced.
#include <cmath>

#include "common.h"

#pragma clang fp contract(off)

namespace psh {
namespace {

constexpr int kSeg = 512;   // samples owned by one thread
constexpr int kWarm = 40;   // recursion warm-up before the owned segment
// (the warm-up is sized for the slowest pole, order 5's 0.4306: |z|^40 = 2e-15 - far below float32 rounding)

__global__ __launch_bounds__(256) void spline_zero_nonfinite(const float *__restrict__ in,
                                                             float *__restrict__ out, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = in[i];
    out[i] = isfinite(v) ? v : 0.f;
  }
}

enum : int { kKindMirror = 0, kKindReflect = 1, kKindWrap = 2 };

// z^k as a float for a negative pole z (0 once it underflows)
__device__ __forceinline__ float pole_pow(float z, int k) {
  return powf(-z, static_cast<float>(k)) * ((k & 1) ? -1.f : 1.f);
}

// zero-padded / edge-padded copy with the missing values zeroed: out (m + 2 npad, n + 2 npad)
__global__ __launch_bounds__(256) void spline_pad(const float *__restrict__ in, float *__restrict__ out, int m, int n,
                                                  int npad, int edge, float cval) {
  const int big_n = n + 2 * npad, big_m = m + 2 * npad;
  const size_t total = static_cast<size_t>(big_m) * big_n;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int y = static_cast<int>(i / big_n) - npad, x = static_cast<int>(i % big_n) - npad;
    const bool inside = y >= 0 && y < m && x >= 0 && x < n;
    float v = cval;
    if (inside || edge) {
      v = in[static_cast<size_t>(min(max(y, 0), m - 1)) * n + min(max(x, 0), n - 1)];
      v = isfinite(v) ? v : 0.f;
    }
    out[i] = v;
  }
}

// causal pass along axis 0 of a (len, width) row-major array: dst[i] = gain src[i] + z dst[i-1]
__global__ __launch_bounds__(64) void spline_causal(const float *__restrict__ src,
                                                    float *__restrict__ dst, int len, int width, int kind, float z,
                                                    float gain) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= width) return;
  const int s0 = blockIdx.y * kSeg, s1 = min(len, s0 + kSeg);
  float c;
  int i;
  if (s0 <= kWarm) {
    // exact mirror initialisation at the array start (ni_splines.c _init_causal_mirror)
    if (len == 1) {
      dst[x] = src[x];  // a single sample is its own coefficient
      return;
    }
    if (kind == kKindMirror) {
      const float zn1 = pole_pow(z, len - 1);
      float acc = gain * src[x] + zn1 * gain * src[static_cast<size_t>(len - 1) * width + x];
      float zi = z;
      const int horizon = min(len - 2, 64);  // |z|^64 ~ 1e-37
      for (int k = 1; k <= horizon; ++k) {
        float term = gain * src[static_cast<size_t>(k) * width + x];
        if (zn1 != 0.f) term += zn1 * gain * src[static_cast<size_t>(len - 1 - k) * width + x];
        acc += zi * term;
        zi *= z;
      }
      c = acc / (1.f - zn1 * zn1);
    } else if (kind == kKindReflect) {  // _init_causal_reflect
      const float zn = pole_pow(z, len), first = gain * src[x];
      float acc = first + zn * gain * src[static_cast<size_t>(len - 1) * width + x];
      float zi = z;
      const int horizon = min(len - 1, 64);
      for (int k = 1; k <= horizon; ++k) {
        float term = gain * src[static_cast<size_t>(k) * width + x];
        if (zn != 0.f) term += zn * gain * src[static_cast<size_t>(len - 1 - k) * width + x];
        acc += zi * term;
        zi *= z;
      }
      c = acc * (z / (1.f - zn * zn)) + first;
    } else {  // _init_causal_wrap
      float acc = gain * src[x];
      float zi = z;
      const int horizon = min(len - 1, 64);
      for (int k = 1; k <= horizon; ++k) {
        acc += zi * gain * src[static_cast<size_t>(len - k) * width + x];
        zi *= z;
      }
      c = acc / (1.f - pole_pow(z, len));
    }
    if (s0 == 0) dst[x] = c;
    i = 1;
  } else {
    i = s0 - kWarm;
    c = gain * src[static_cast<size_t>(i) * width + x];  // any start value: forgotten after kWarm steps
    ++i;
  }
  for (; i < s1; ++i) {
    c = gain * src[static_cast<size_t>(i) * width + x] + z * c;
    if (i >= s0) dst[static_cast<size_t>(i) * width + x] = c;
  }
}

// anticausal pass: dst[i] = z (dst[i+1] - cp[i]) with cp the causal result
__global__ __launch_bounds__(64) void spline_anticausal(const float *__restrict__ cp,
                                                        float *__restrict__ dst, int len,
                                                        int width, int kind, float z) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  if (x >= width || len == 1) {
    if (x < width && blockIdx.y == 0) dst[x] = cp[x];
    return;
  }
  const int s0 = blockIdx.y * kSeg, s1 = min(len, s0 + kSeg);
  float c;
  int i;
  if (s1 + kWarm >= len) {
    // exact mirror initialisation at the array end (_init_anticausal_mirror)
    const float last = cp[static_cast<size_t>(len - 1) * width + x];
    if (kind == kKindMirror) {
      const float prev = cp[static_cast<size_t>(len - 2) * width + x];
      c = (z * prev + last) * z / (z * z - 1.f);
    } else if (kind == kKindReflect) {  // _init_anticausal_reflect
      c = last * (z / (z - 1.f));
    } else {  // _init_anticausal_wrap: the causal values at the START of the array
      float acc = last;
      float zi = z;
      const int horizon = min(len - 1, 64);
      for (int k = 0; k < horizon; ++k) {
        acc += zi * cp[static_cast<size_t>(k) * width + x];
        zi *= z;
      }
      c = acc * (z / (pole_pow(z, len) - 1.f));
    }
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

// coef <- B-spline coefficients (order 2 .. 5) of precip (NaN/Inf -> 0) padded by `npad` samples (edge values
// or `cval`), boundary kind 0 mirror / 1 reflect / 2 wrap; coef and tmp: (m + 2 npad, n + 2 npad) planes.
hipError_t spline_prefilter(const float *precip, float *coef, float *tmp, int m, int n, hipStream_t stream, int kind,
                            int npad, int pad_edge, float cval, int order) {
  double poles[2] = {std::sqrt(3.0) - 2.0, 0.0};
  int npoles = 1;
  if (order == 2) {
    poles[0] = std::sqrt(8.0) - 3.0;
  } else if (order == 4) {
    poles[0] = std::sqrt(664.0 - std::sqrt(438976.0)) + std::sqrt(304.0) - 19.0;
    poles[1] = std::sqrt(664.0 + std::sqrt(438976.0)) - std::sqrt(304.0) - 19.0;
    npoles = 2;
  } else if (order == 5) {
    poles[0] = std::sqrt(67.5 - std::sqrt(4436.25)) + std::sqrt(26.25) - 6.5;
    poles[1] = std::sqrt(67.5 + std::sqrt(4436.25)) - std::sqrt(26.25) - 6.5;
    npoles = 2;
  }
  double gain = 1.0;
  for (int k = 0; k < npoles; ++k) gain *= (1.0 - poles[k]) * (1.0 - 1.0 / poles[k]);
  if (npad > 0) {
    hipLaunchKernelGGL(spline_pad, dim3(2048), dim3(256), 0, stream, precip, coef, m, n, npad, pad_edge, cval);
    m += 2 * npad;
    n += 2 * npad;
  } else {
    hipLaunchKernelGGL(spline_zero_nonfinite, dim3(2048), dim3(256), 0, stream, precip, coef, static_cast<size_t>(m) * n);
  }
  // axis 0 (columns of the image): one causal / anticausal pair per pole, coef -> tmp -> coef
  dim3 g0((n + 63) / 64, (m + kSeg - 1) / kSeg);
  for (int k = 0; k < npoles; ++k) {
    const float z = static_cast<float>(poles[k]);
    hipLaunchKernelGGL(spline_causal, g0, dim3(64), 0, stream, coef, tmp, m, n, kind, z, k == 0 ? static_cast<float>(gain) : 1.f);
    hipLaunchKernelGGL(spline_anticausal, g0, dim3(64), 0, stream, tmp, coef, m, n, kind, z);
  }
  // axis 1 (rows) as columns of the transpose: tmp -> coef -> tmp per pole
  hipLaunchKernelGGL(transpose32, dim3((n + 31) / 32, (m + 31) / 32), dim3(256), 0, stream, coef, tmp, m, n);
  dim3 g1((m + 63) / 64, (n + kSeg - 1) / kSeg);
  for (int k = 0; k < npoles; ++k) {
    const float z = static_cast<float>(poles[k]);
    hipLaunchKernelGGL(spline_causal, g1, dim3(64), 0, stream, tmp, coef, n, m, kind, z, k == 0 ? static_cast<float>(gain) : 1.f);
    hipLaunchKernelGGL(spline_anticausal, g1, dim3(64), 0, stream, coef, tmp, n, m, kind, z);
  }
  hipLaunchKernelGGL(transpose32, dim3((m + 31) / 32, (n + 31) / 32), dim3(256), 0, stream, tmp, coef, n, m);
  return hipGetLastError();
}

}  // namespace psh
