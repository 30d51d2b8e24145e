// Evaluation of a radial-basis-function interpolant on the full pixel grid, for gfx950.
//
// pysteps/utils/interpolate.py:117-170 (rbfinterp2d) wraps scipy.interpolate.Rbf [third party, installed]:
// the weights come from ONE dense N x N solve (N = number of sparse vectors, a few hundred to a few
// thousand: host LAPACK through SciPy itself, so they are the reference's weights bit for bit), the
// interpolant  f(x) = sum_j w_j phi(|x - x_j|)  is then evaluated at every grid node - N x m x n basis
// function values, 1.7e10 at 4096^2 with 1000 vectors, which is where the reference spends its time
// (a (chunk pixels) x N distance matrix per grid chunk in NumPy).  That evaluation is this kernel.
//
// float64 throughout (the weights of an RBF system alternate in sign and can be large: float32 sums
// would cancel).  A thread owns four pixels of a column (rows y, y + 4, y + 8, y + 12 of a 64 x 16
// tile: 64-lane rows are written coalesced), the nodes stream through LDS in chunks of 256
// (x, y, w_u, w_v), every node is read once per four pixels.  fp64 add / fma issue at the fp32 rate on
// this chip (4 cycles per wave instruction), the square root is the expensive part.
#include "common.h"

namespace psh {
namespace {

enum RbfFunction : int { kMultiquadric = 0, kInverse, kGaussian, kLinear, kCubic, kQuintic, kThinPlate, kNumRbf };

constexpr int kRbfChunk = 256;  // nodes staged in LDS at a time
constexpr int kRbfPx = 4;       // pixels per thread

template <int FN>
__device__ __forceinline__ double rbf_phi(double r2, double inv_eps2) {
  // scipy.interpolate.Rbf._h_*: r is the Euclidean distance, epsilon the shape parameter
  if constexpr (FN == kMultiquadric) return sqrt(r2 * inv_eps2 + 1.0);        // sqrt((r / eps)^2 + 1)
  if constexpr (FN == kInverse) return 1.0 / sqrt(r2 * inv_eps2 + 1.0);       // 1 / sqrt((r / eps)^2 + 1)
  if constexpr (FN == kGaussian) return exp(-(r2 * inv_eps2));                // exp(-(r / eps)^2)
  if constexpr (FN == kLinear) return sqrt(r2);                               // r
  if constexpr (FN == kCubic) return r2 * sqrt(r2);                           // r^3
  if constexpr (FN == kQuintic) return r2 * r2 * sqrt(r2);                    // r^5
  return r2 > 0.0 ? 0.5 * r2 * log(r2) : 0.0;                                 // thin plate: r^2 log r (xlogy: 0 at 0)
}

template <int FN>
__global__ __launch_bounds__(256) void rbf_eval(const double2 *__restrict__ xy, const double2 *__restrict__ w, int N,
                                                int m, int n, double x0, double dx, double y0, double dy, double inv_eps2,
                                                double *__restrict__ out) {
  __shared__ double4 s_node[kRbfChunk];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ix = blockIdx.x * 64 + lane;
  const int iy0 = blockIdx.y * (4 * kRbfPx) + wave;  // rows iy0 + 4 q
  const double px = x0 + dx * static_cast<double>(ix);
  double py[kRbfPx], su[kRbfPx], sv[kRbfPx];
#pragma unroll
  for (int q = 0; q < kRbfPx; ++q) {
    py[q] = y0 + dy * static_cast<double>(iy0 + 4 * q);
    su[q] = sv[q] = 0.0;
  }
  for (int j0 = 0; j0 < N; j0 += kRbfChunk) {
    __syncthreads();  // the previous chunk's readers are done
    const int j = j0 + threadIdx.x;
    if (j < N) {
      const double2 p = xy[j], v = w[j];
      s_node[threadIdx.x] = make_double4(p.x, p.y, v.x, v.y);
    }
    __syncthreads();
    const int cnt = min(kRbfChunk, N - j0);
    for (int t = 0; t < cnt; ++t) {
      const double4 nd = s_node[t];  // same address in every lane: LDS broadcast
      const double ddx = px - nd.x, ddx2 = ddx * ddx;
#pragma unroll
      for (int q = 0; q < kRbfPx; ++q) {
        const double ddy = py[q] - nd.y;
        const double phi = rbf_phi<FN>(fma(ddy, ddy, ddx2), inv_eps2);
        su[q] = fma(nd.z, phi, su[q]);
        sv[q] = fma(nd.w, phi, sv[q]);
      }
    }
  }
  if (ix >= n) return;
  const size_t plane = static_cast<size_t>(m) * n;
#pragma unroll
  for (int q = 0; q < kRbfPx; ++q) {
    const int iy = iy0 + 4 * q;
    if (iy < m) {
      out[static_cast<size_t>(iy) * n + ix] = su[q];
      out[plane + static_cast<size_t>(iy) * n + ix] = sv[q];
    }
  }
}

}  // namespace
}  // namespace psh

// xy_dev: (N, 2) node coordinates, weights_dev: (N, 2) weights of two variables (zeros for a missing second
// one), both float64; out_dev: (2, m, n) float64 on the regular grid x = x0 + dx i, y = y0 + dy j.
// function: 0 multiquadric, 1 inverse, 2 gaussian, 3 linear, 4 cubic, 5 quintic, 6 thin_plate (the names of
// scipy.interpolate.Rbf); epsilon its shape parameter.  Asynchronous on the library stream.
extern "C" int psh_rbf_eval_dev(const double *xy_dev, const double *weights_dev, int N, int m, int n, double x0, double dx,
                                double y0, double dy, int function, double epsilon, double *out_dev) {
  using namespace psh;
  PSH_REQUIRE_INIT();
  if (!xy_dev || !weights_dev || !out_dev) return fail(PSH_EINVAL, "rbf_eval: NULL pointer");
  if (N <= 0 || m <= 0 || n <= 0) return fail(PSH_EINVAL, "rbf_eval: invalid sizes (N=%d, grid %d x %d)", N, m, n);
  if (function < 0 || function >= kNumRbf) return fail(PSH_EUNSUPPORTED, "rbf_eval: basis function %d not implemented", function);
  if (!(epsilon > 0.0) && function <= kGaussian) return fail(PSH_EINVAL, "rbf_eval: epsilon must be positive");
  Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const dim3 grid((n + 63) / 64, (m + 4 * kRbfPx - 1) / (4 * kRbfPx)), block(256);
  const double inv_eps2 = function <= kGaussian ? 1.0 / (epsilon * epsilon) : 0.0;
  const double2 *xy = reinterpret_cast<const double2 *>(xy_dev), *w = reinterpret_cast<const double2 *>(weights_dev);
#define PSH_RBF(FN) \
  case FN: hipLaunchKernelGGL(rbf_eval<FN>, grid, block, 0, c.stream, xy, w, N, m, n, x0, dx, y0, dy, inv_eps2, out_dev); break;
  switch (function) {
    PSH_RBF(0) PSH_RBF(1) PSH_RBF(2) PSH_RBF(3) PSH_RBF(4) PSH_RBF(5) PSH_RBF(6)
  }
#undef PSH_RBF
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}
