// Local multivariate outlier test of sparse motion vectors on the device.
//
// Replaces the form of pysteps/utils/cleansing.py:124-249 (detect_outliers) that
// dense_lucaskanade uses (pysteps/motion/lucaskanade.py:252: uv (N,2), coord xy
// (N,2), k nearest neighbours): for every vector the k nearest OTHER vectors
// (cKDTree k+1 query minus the first hit, :221-224), z = uv - mean(neighbours),
// C = cov(neighbours, ddof=1), outlier iff sqrt(z^T C^-1 z) > thr; singular C ->
// not an outlier (:239-243).
//
// N is a few thousand at most: one thread owns one vector, scans all N in LDS-
// staged chunks and keeps its k+1 nearest in registers; everything is float64
// (coordinates are exact, ties are broken by the lower index - cKDTree's own tie
// order is unspecified).  Runtime ~ tens of microseconds; it exists to keep the
// 2-4 ms host k-d tree query off the critical path of a nowcast step.
#include <vector>

#include "common.h"

namespace psh {
namespace {

constexpr int kQcMaxK = 64;   // k + 1 <= 64
constexpr int kQcChunk = 256;

template <int KMAX>
__global__ __launch_bounds__(64) void outliers_local(const double2 *__restrict__ xy,
                                                     const double2 *__restrict__ uv, int n, int k,
                                                     double thr,
                                                     unsigned char *__restrict__ flags) {
  __shared__ double2 s_xy[kQcChunk];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const double2 me = xy[live ? i : 0];
  const int kk = min(n, k + 1);  // neighbours incl. the vector itself
  double d2[KMAX];
  int idx[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    d2[j] = j < kk ? INFINITY : -INFINITY;
    idx[j] = -1;
  }
  double worst = INFINITY;
  int worst_pos = 0;
  for (int base = 0; base < n; base += kQcChunk) {
    __syncthreads();
    for (int t = threadIdx.x; t < kQcChunk && base + t < n; t += blockDim.x) s_xy[t] = xy[base + t];
    __syncthreads();
    const int lim = min(kQcChunk, n - base);
    for (int t = 0; t < lim; ++t) {
      const double dx = s_xy[t].x - me.x, dy = s_xy[t].y - me.y;
      const double d = dx * dx + dy * dy;
      if (d < worst) {  // strict: among equal distances the lower index stays
        double w = -INFINITY;
        int wp = 0;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          const bool hit = j == worst_pos;
          d2[j] = hit ? d : d2[j];
          idx[j] = hit ? base + t : idx[j];
          if (d2[j] > w) {
            w = d2[j];
            wp = j;
          }
        }
        worst = w;
        worst_pos = wp;
      }
    }
  }
  if (!live) return;
  // drop the nearest hit (the vector itself, or a duplicate position with a lower index)
  double best = INFINITY;
  int best_idx = 0x7fffffff, best_pos = 0;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (idx[j] >= 0 && (d2[j] < best || (d2[j] == best && idx[j] < best_idx))) {
      best = d2[j];
      best_idx = idx[j];
      best_pos = j;
    }
  }
  double su = 0.0, sv = 0.0;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    if (idx[j] >= 0 && j != best_pos) {
      const double2 q = uv[idx[j]];
      su += q.x;
      sv += q.y;
      ++cnt;
    }
  }
  bool out = false;
  if (cnt >= 2) {
    const double mu = su / cnt, mv = sv / cnt;
    double cuu = 0.0, cuv = 0.0, cvv = 0.0;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (idx[j] >= 0 && j != best_pos) {
        const double2 q = uv[idx[j]];
        const double a = q.x - mu, b = q.y - mv;
        cuu += a * a;
        cuv += a * b;
        cvv += b * b;
      }
    }
    const double dof = cnt - 1;
    cuu /= dof;
    cuv /= dof;
    cvv /= dof;
    const double det = cuu * cvv - cuv * cuv;
    if (det != 0.0 && isfinite(det)) {
      const double2 mine = uv[i];
      const double zu = mine.x - mu, zv = mine.y - mv;
      const double md2 = (zu * zu * cvv - 2.0 * zu * zv * cuv + zv * zv * cuu) / det;
      out = sqrt(md2) > thr;
    }
  }
  flags[i] = out ? 1 : 0;
}

}  // namespace
}  // namespace psh

extern "C" int psh_outliers_local_host(const double *xy, const double *values, int n, int k,
                                       double thr, unsigned char *flags) {
  PSH_REQUIRE_INIT();
  if (n < 0) return psh::fail(PSH_EINVAL, "outliers: negative sample count");
  if (n == 0) return PSH_OK;
  if (!xy || !values || !flags) return psh::fail(PSH_EINVAL, "outliers: NULL pointer");
  if (k < 1) return psh::fail(PSH_EINVAL, "outliers: k must be >= 1");
  if (k + 1 > psh::kQcMaxK && n > psh::kQcMaxK)
    return psh::fail(PSH_EUNSUPPORTED, "outliers: k=%d > %d not implemented on the device", k,
                     psh::kQcMaxK - 1);
  if (n < 2) {
    flags[0] = 0;
    return PSH_OK;
  }
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t vec_bytes = static_cast<size_t>(n) * 2 * sizeof(double);
  void *blk = nullptr;
  if (int rc = psh_malloc(&blk, 2 * vec_bytes + static_cast<size_t>(n))) return rc;
  char *base = static_cast<char *>(blk);
  double2 *d_xy = reinterpret_cast<double2 *>(base);
  double2 *d_uv = reinterpret_cast<double2 *>(base + vec_bytes);
  unsigned char *d_fl = reinterpret_cast<unsigned char *>(base + 2 * vec_bytes);
  int rc = PSH_OK;
  auto run = [&]() -> int {
    PSH_HIP(hipMemcpyAsync(d_xy, xy, vec_bytes, hipMemcpyHostToDevice, c.stream));
    PSH_HIP(hipMemcpyAsync(d_uv, values, vec_bytes, hipMemcpyHostToDevice, c.stream));
    const dim3 grid((n + 63) / 64), block(64);
    if (std::min(n, k + 1) <= 32) {
      hipLaunchKernelGGL((psh::outliers_local<32>), grid, block, 0, c.stream, d_xy, d_uv, n, k, thr, d_fl);
    } else {
      hipLaunchKernelGGL((psh::outliers_local<psh::kQcMaxK>), grid, block, 0, c.stream, d_xy, d_uv, n, k, thr, d_fl);
    }
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(flags, d_fl, static_cast<size_t>(n), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    return PSH_OK;
  };
  rc = run();
  (void)psh_free(blk);
  return rc;
}
