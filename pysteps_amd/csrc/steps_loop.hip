// The element-wise half of one STEPS member update on the device (the loop itself):
// what pysteps/nowcasts/steps.py does per member and time step in `__update_state` (:1057-1219)
// between the spectral operators (noise filter, cascade decomposition: cascade.hip), the CDF matching
// (probmatch.hip) and the incremental mask (mask.hip).  Everything float64 like the reference, every
// product and sum rounded on its own (NumPy evaluates each line as separate array operations), so the
// results are bit-identical with the reference's arithmetic.
//
//  * psh_steps_ar_recompose_dev - :1116-1146 + :1176-1185, fused over all cascade levels of a member:
//      eps_k *= noise_std_coeffs[k]                                           (:1131-1132)
//      x_new,k = 0.0 + phi_k1 x_k[-1] + ... + phi_kp x_k[-p] + phi_k,p+1 eps_k   (autoregression.py:1056-1070)
//      field = sum_k (x_new,k sigma_k + mu_k)                                 (decomposition.py:294-301)
//    The AR history of a level is a ring of p planes: x_new overwrites the oldest one (the reference
//    copies the series up by one, `x[1:]` + new).  Optionally the field's minimum (np.min) lands in a
//    device word for the masking step.  Moves (p + 2) L + 1 planes once; no level ever leaves HBM.
//  * psh_steps_mask_dev         - :1221-1240 `__apply_precipitation_mask` (incremental: grey-scale mask;
//      obs: boolean mask): pf = min + (pf - min) mask, pixels not above the minimum set to it
//  * psh_steps_mean_shift_dev   - :1203-1206 probmatching_method="mean"
//  * psh_ge_mask_dev, psh_nan_where_dev, psh_lerp_dev - `precip_forecast >= precip_thr` (:1211),
//      `precip_forecast[domain_mask] = nan` (:1217), `(1 - w) prev + w new` (nowcasts/utils.py:419-427)
#include <algorithm>

#include "common.h"

namespace psh {
namespace {

constexpr int kMaxLevels = 16, kMaxOrder = 8, kThreads = 256;

struct ArRecompose {
  double phi[kMaxLevels][kMaxOrder + 1];
  double eps_scale[kMaxLevels], mu[kMaxLevels], sigma[kMaxLevels];
  int nlevels, p, head, has_eps;
};

// order-preserving map double -> uint64 (NaN -> 0, the smallest key: np.min propagates NaN)
__device__ __forceinline__ unsigned long long min_key(double v) {
  if (v != v) return 0ull;
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_min_key(unsigned long long k) {
  if (k == 0ull) return __longlong_as_double(0x7ff8000000000000ll);
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double(static_cast<long long>(b));
}

__device__ __forceinline__ void publish_min(unsigned long long key, unsigned long long *dst) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = __shfl_xor(key, d);
    key = o < key ? o : key;
  }
  if ((threadIdx.x & 63) == 0 && key < __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(dst, key);
}

// cascades: (L, p, plane) of one member; eps: (L, plane) or nullptr; one thread = two pixels
// eps_stats (nullable): (mean, std) per level of `eps` - the decomposition left its levels as they came
// out of the transforms and the standardisation (x - mean) / std of decomposition.py:224-232 is applied
// here, to the value on its way in: the same two operations, one sweep over the levels less
__global__ __launch_bounds__(kThreads) void ar_recompose(double *__restrict__ cascades, const double *__restrict__ eps,
                                                         const double2 *__restrict__ eps_stats, size_t plane, ArRecompose a,
                                                         double *__restrict__ field, unsigned long long *__restrict__ min_out) {
#pragma clang fp contract(off)
  const size_t pairs = plane / 2;
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  unsigned long long key = ~0ull;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < pairs; i += stride) {
    double2 total = make_double2(0.0, 0.0);
    for (int k = 0; k < a.nlevels; ++k) {
      double *lvl = cascades + static_cast<size_t>(k) * a.p * plane;
      double2 acc = make_double2(0.0, 0.0);
      for (int j = 0; j < a.p; ++j) {  // x[-1-j] sits in slot (head + p - 1 - j) mod p
        int slot = a.head + a.p - 1 - j;
        if (slot >= a.p) slot -= a.p;
        const double2 x = reinterpret_cast<const double2 *>(lvl + static_cast<size_t>(slot) * plane)[i];
        const double tx = a.phi[k][j] * x.x, ty = a.phi[k][j] * x.y;
        acc.x = acc.x + tx;
        acc.y = acc.y + ty;
      }
      if (a.has_eps) {
        double2 e = reinterpret_cast<const double2 *>(eps + static_cast<size_t>(k) * plane)[i];
        if (eps_stats) {
          const double2 st = eps_stats[k];
          const double cx = e.x - st.x, cy = e.y - st.x;
          e.x = cx / st.y;
          e.y = cy / st.y;
        }
        const double ex = e.x * a.eps_scale[k], ey = e.y * a.eps_scale[k];
        const double tx = a.phi[k][a.p] * ex, ty = a.phi[k][a.p] * ey;
        acc.x = acc.x + tx;
        acc.y = acc.y + ty;
      }
      reinterpret_cast<double2 *>(lvl + static_cast<size_t>(a.head) * plane)[i] = acc;
      const double sx = acc.x * a.sigma[k], sy = acc.y * a.sigma[k];
      const double vx = sx + a.mu[k], vy = sy + a.mu[k];
      total.x = k == 0 ? vx : total.x + vx;
      total.y = k == 0 ? vy : total.y + vy;
    }
    reinterpret_cast<double2 *>(field)[i] = total;
    const unsigned long long kx = min_key(total.x), ky = min_key(total.y);
    key = kx < key ? kx : key;
    key = ky < key ? ky : key;
  }
  if ((plane & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd plane: the last pixel on its own
    const size_t i = plane - 1;
    double total = 0.0;
    for (int k = 0; k < a.nlevels; ++k) {
      double *lvl = cascades + static_cast<size_t>(k) * a.p * plane;
      double acc = 0.0;
      for (int j = 0; j < a.p; ++j) {
        int slot = a.head + a.p - 1 - j;
        if (slot >= a.p) slot -= a.p;
        const double t = a.phi[k][j] * lvl[static_cast<size_t>(slot) * plane + i];
        acc = acc + t;
      }
      if (a.has_eps) {
        double e0 = eps[static_cast<size_t>(k) * plane + i];
        if (eps_stats) {
          const double c0 = e0 - eps_stats[k].x;
          e0 = c0 / eps_stats[k].y;
        }
        const double e = e0 * a.eps_scale[k];
        const double t = a.phi[k][a.p] * e;
        acc = acc + t;
      }
      lvl[static_cast<size_t>(a.head) * plane + i] = acc;
      const double s = acc * a.sigma[k];
      const double v = s + a.mu[k];
      total = k == 0 ? v : total + v;
    }
    field[i] = total;
    const unsigned long long kk = min_key(total);
    key = kk < key ? kk : key;
  }
  if (min_out) publish_min(key, min_out);
}

// steps.py:1221-1240
__global__ __launch_bounds__(kThreads) void apply_mask(double *__restrict__ field, size_t n, const double *__restrict__ grey,
                                                       const unsigned char *__restrict__ keep,
                                                       const unsigned long long *__restrict__ min_key_in) {
#pragma clang fp contract(off)
  const double mn = from_min_key(*min_key_in);
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
    double v = field[i];
    if (grey) {
      const double d = v - mn;
      const double s = d * grey[i];
      v = mn + s;
      if (!(v > mn)) v = mn;
    } else if (!keep[i]) {
      v = mn;
    }
    field[i] = v;
  }
}

__global__ __launch_bounds__(kThreads) void ge_mask(const double *__restrict__ field, size_t n, double thr,
                                                    unsigned char *__restrict__ out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) out[i] = field[i] >= thr ? 1 : 0;
}

__global__ __launch_bounds__(kThreads) void nan_where(double *__restrict__ field, const unsigned char *__restrict__ mask, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride)
    if (mask[i]) field[i] = __longlong_as_double(0x7ff8000000000000ll);
}

__global__ __launch_bounds__(kThreads) void lerp(const double *__restrict__ a, const double *__restrict__ b, double w,
                                                 double *__restrict__ out, size_t n) {
#pragma clang fp contract(off)
  const double wa = 1.0 - w;
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
    const double ta = wa * a[i], tb = w * b[i];
    out[i] = ta + tb;
  }
}

// probmatching_method="mean" (steps.py:1203-1206): sum and count of the values >= thr, then
// v - mean + mu_0 on them.  The sum is a tree of partial sums, not NumPy's pairwise order: the mean
// agrees with np.mean to ~1e-16 relative.
__global__ __launch_bounds__(kThreads) void wet_sum(const double *__restrict__ field, size_t n, double thr,
                                                    double2 *__restrict__ partial) {
  __shared__ double2 s_part[kThreads / 64];
  double s = 0.0, c = 0.0;
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
    const double v = field[i];
    if (v >= thr) {
      s += v;
      c += 1.0;
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    s += __shfl_xor(s, d);
    c += __shfl_xor(c, d);
  }
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = make_double2(s, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    double2 t = s_part[0];
    for (int w = 1; w < kThreads / 64; ++w) {
      t.x += s_part[w].x;
      t.y += s_part[w].y;
    }
    partial[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kThreads) void mean_shift(double *__restrict__ field, size_t n, double thr, double mu0,
                                                       const double2 *__restrict__ partial, int nparts) {
#pragma clang fp contract(off)
  __shared__ double s_mean;
  if (threadIdx.x == 0) {  // every block sums the partials in the same order
    double s = 0.0, c = 0.0;
    for (int k = 0; k < nparts; ++k) {
      s += partial[k].x;
      c += partial[k].y;
    }
    s_mean = s / c;  // no wet pixel: 0 / 0 = NaN, nothing is touched below (np.mean of an empty array)
  }
  __syncthreads();
  const double mean = s_mean;
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
    const double v = field[i];
    if (v >= thr) {
      const double d = v - mean;
      field[i] = d + mu0;
    }
  }
}

unsigned grid_for(size_t n) {
  const size_t blocks = (n + kThreads - 1) / kThreads;
  return static_cast<unsigned>(blocks < 8192 ? (blocks ? blocks : 1) : 8192);
}

// ---- the member update in the spectral domain ---------------------------------------------------------
// Everything between the white noise and the recomposed field is linear (noise filter, band-pass
// decomposition, AR(p) step, recomposition) except the two standardisations - of the noise field and of
// its cascade levels (fftgenerators.py:433-437, decomposition.py:217-232) - and those need only second
// moments, which Parseval's identity reads off the spectrum.  So the AR history is kept as SPECTRA
// (rfft2 of the reference's level fields) and a member update needs TWO transforms - rfft2 of the white
// noise, irfft2 of the recomposed spectrum - instead of nine:
//   y     = rfft2(white) F                           (the noise field's spectrum; its mean is removed = DC -> 0)
//   B_k   = sum' |y w_k|^2                            (Hermitian-weighted over the half spectrum, DC excluded)
//   std(noise) std(level k) = sqrt(B_k) / (m n)       (the noise field's own variance cancels)
//   X_k  <- phi_k1 X_k[-1] + ... + phi_kp X_k[-p] + phi_k,p+1 noise_std_k (m n) / sqrt(B_k) y w_k
//   field = irfft2( sum_k sigma_k X_k  +  (sum_k mu_k) m n at DC )
// The same numbers as the chain of the reference's spatial operators up to rounding (every step of
// that chain goes through transforms already); this is also what the reference's own domain="spectral"
// option does (steps.py:122-126).  Band-pass weights and noise filter are real and symmetric (functions
// of |k|), so every spectrum stays Hermitian.
struct SpectralAr {
  double phi[kMaxLevels][kMaxOrder + 1];
  double gain[kMaxLevels];  // phi_k,p+1 noise_std_k m n        (divided by sqrt(B_k) in the kernel)
  double sigma[kMaxLevels];
  double dc_add;            // (sum_k mu_k) m n
  int nlevels, p, head;
};

// hermitian weight of column c of an rfft2 half spectrum: interior columns stand for two coefficients
__device__ __forceinline__ double herm_weight(int c, int nc, int n_even) { return (c == 0 || (n_even && c == nc - 1)) ? 1.0 : 2.0; }

__global__ __launch_bounds__(kThreads) void spectral_level_sums(const double2 *__restrict__ noise, const double *__restrict__ filt,
                                                                const double *__restrict__ weights, int nlevels, int m,
                                                                int nc, int n_even, double *__restrict__ partial) {
  __shared__ double s_part[kThreads / 64][kMaxLevels];
  const size_t plane = static_cast<size_t>(m) * nc;
  double acc[kMaxLevels];
#pragma unroll
  for (int k = 0; k < kMaxLevels; ++k) acc[k] = 0.0;
  for (int r = blockIdx.x; r < m; r += gridDim.x) {  // whole rows per workgroup: the column (its Hermitian weight) without a division
    for (int c = threadIdx.x; c < nc; c += kThreads) {
      if (r == 0 && c == 0) continue;  // DC: the noise field's mean, removed by its standardisation
      const size_t i = static_cast<size_t>(r) * nc + c;
      const double2 y = noise[i];
      const double f = filt[i];
      const double e = herm_weight(c, nc, n_even) * f * f * (y.x * y.x + y.y * y.y);
#pragma unroll
      for (int k = 0; k < kMaxLevels; ++k) {
        if (k < nlevels) {
          const double w = weights[static_cast<size_t>(k) * plane + i];
          acc[k] += e * w * w;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kMaxLevels; ++k) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc[k] += __shfl_xor(acc[k], d);
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < kMaxLevels; ++k) s_part[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < kMaxLevels) {
    double t = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) t += s_part[w][threadIdx.x];
    partial[static_cast<size_t>(blockIdx.x) * kMaxLevels + threadIdx.x] = t;
  }
}

// one workgroup per level adds the workgroups' partial sums
__global__ __launch_bounds__(kThreads) void spectral_level_sums_final(const double *__restrict__ partial, int nparts,
                                                                      double *__restrict__ sums) {
  __shared__ double s_part[kThreads / 64];
  const int k = blockIdx.x;
  double t = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kThreads) t += partial[static_cast<size_t>(i) * kMaxLevels + k];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    double all = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) all += s_part[w];
    sums[k] = all;
  }
}

// cascades: (L, p, plane) complex rings of one member; writes the new spectra into slot `head` and the
// recomposed spectrum into `out`
__global__ __launch_bounds__(kThreads) void spectral_ar(double2 *__restrict__ cascades, const double2 *__restrict__ noise,
                                                        const double *__restrict__ filt, const double *__restrict__ weights,
                                                        const double *__restrict__ sums, size_t plane, SpectralAr a,
                                                        double2 *__restrict__ out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  __shared__ double gain[kMaxLevels];
  if (threadIdx.x < kMaxLevels) gain[threadIdx.x] = static_cast<int>(threadIdx.x) < a.nlevels ? a.gain[threadIdx.x] / sqrt(sums[threadIdx.x]) : 0.0;
  __syncthreads();
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < plane; i += stride) {
    const double2 y0 = noise[i];
    const double f = i == 0 ? 0.0 : filt[i];  // the standardised noise has no DC
    const double2 y = make_double2(y0.x * f, y0.y * f);
    double2 total = make_double2(i == 0 ? a.dc_add : 0.0, 0.0);
    for (int k = 0; k < a.nlevels; ++k) {
      double2 *lvl = cascades + static_cast<size_t>(k) * a.p * plane;
      const double g = gain[k] * weights[static_cast<size_t>(k) * plane + i];
      double2 acc = make_double2(g * y.x, g * y.y);
      for (int j = 0; j < a.p; ++j) {  // x[-1-j] sits in slot (head + p - 1 - j) mod p
        int slot = a.head + a.p - 1 - j;
        if (slot >= a.p) slot -= a.p;
        const double2 x = lvl[static_cast<size_t>(slot) * plane + i];
        acc.x += a.phi[k][j] * x.x;
        acc.y += a.phi[k][j] * x.y;
      }
      lvl[static_cast<size_t>(a.head) * plane + i] = acc;
      total.x += a.sigma[k] * acc.x;
      total.y += a.sigma[k] * acc.y;
    }
    out[i] = total;
  }
}

// ---- the reference's own domain="spectral" (steps.py:122-126, 1150-1166) ----------------------------------
// There the state IS spectral: the AR history of level k lives on the coefficients where the band-pass
// weight exceeds 1e-12 ("compact" arrays), the noise is a field of unit phasors exp(i theta) with theta drawn
// by RandomState.uniform (fftgenerators.py:407-418), filtered, standardised and decomposed without a
// transform, and one irfft2 per member update turns the recomposed spectrum into the field.  Here the history
// is kept as full half-spectrum planes (entries outside a level's mask are never read), one kernel does
// noise -> levels -> AR(p) -> recomposition per coefficient:
//   N = cos(theta) + i sin(theta)  (column 0 mirrored: theta[m - r, 0] = -theta[r, 0]);  y = N F, y[0, 0] = 0
//   y /= std(y)                      (a complex array divided by a real scalar: NumPy multiplies by the reciprocal)
//   e_k = (y W_k - mean_k) / std_k   (mean_k = 0: y has no DC;  std_k by Parseval - both standard deviations only
//                                     depend on |N| = 1, F and W_k: constants of the nowcast, computed once on the host)
//   x_k <- phi_k1 x_k[-1] + ... + phi_kp x_k[-p] + phi_k,p+1 (noise_std_k e_k)        on mask_k = {W_k > 1e-12}
//   R   = sum_k mask_k (sigma_k x_k + mu_k)
// The arithmetic follows the reference's sequence of element-wise operations; cos / sin come from the device
// library (NumPy's from libm): parity to rounding, not bit for bit.
struct PhaseAr {
  double phi[kMaxLevels][kMaxOrder + 1];
  double inv_std[kMaxLevels];  // 1 / std_k
  double noise_std[kMaxLevels];
  double mu[kMaxLevels], sigma[kMaxLevels];
  double inv_stdn;             // 1 / std(y)
  int nlevels, p, head, m, nc;
};

__global__ __launch_bounds__(kThreads) void spectral_phase_ar(double2 *__restrict__ cascades, const double *__restrict__ theta,
                                                              const double *__restrict__ filt, const double *__restrict__ weights,
                                                              PhaseAr a, double2 *__restrict__ out) {
#pragma clang fp contract(off)
  const size_t plane = static_cast<size_t>(a.m) * a.nc;
  const int first_mirrored = a.m / 2 + 1;  // rows of column 0 that repeat an earlier row's phase with the opposite sign
  for (int r = blockIdx.x; r < a.m; r += gridDim.x) {
    for (int c = threadIdx.x; c < a.nc; c += kThreads) {
      const size_t i = static_cast<size_t>(r) * a.nc + c;
      double2 y = make_double2(0.0, 0.0);
      if (theta) {  // (uniform) nullptr: the deterministic model of the S-PROG mask, no innovation term (steps.py:1089-1097)
        const bool mirrored = c == 0 && r >= first_mirrored;
        const double t = mirrored ? -theta[static_cast<size_t>(a.m - r) * a.nc] : theta[i];
        double sn, cs;
        sincos(t, &sn, &cs);
        const double f = filt[i];
        if (i != 0) y = make_double2(cs * f, sn * f);
        y.x *= a.inv_stdn;
        y.y *= a.inv_stdn;
      }
      double2 total = make_double2(0.0, 0.0);
      for (int k = 0; k < a.nlevels; ++k) {
        const double w = weights[static_cast<size_t>(k) * plane + i];
        if (!(w > 1e-12)) continue;  // not a coefficient of this level
        double2 e = make_double2(y.x * w, y.y * w);
        e.x *= a.inv_std[k];
        e.y *= a.inv_std[k];
        e.x *= a.noise_std[k];
        e.y *= a.noise_std[k];
        double2 *lvl = cascades + static_cast<size_t>(k) * a.p * plane;
        double2 acc = make_double2(0.0, 0.0);
        for (int j = 0; j < a.p; ++j) {  // x[-1-j] sits in slot (head + p - 1 - j) mod p
          int slot = a.head + a.p - 1 - j;
          if (slot >= a.p) slot -= a.p;
          const double2 x = lvl[static_cast<size_t>(slot) * plane + i];
          acc.x += a.phi[k][j] * x.x;
          acc.y += a.phi[k][j] * x.y;
        }
        if (theta) {
          acc.x += a.phi[k][a.p] * e.x;
          acc.y += a.phi[k][a.p] * e.y;
        }
        lvl[static_cast<size_t>(a.head) * plane + i] = acc;
        total.x += acc.x * a.sigma[k] + a.mu[k];
        total.y += acc.y * a.sigma[k];
      }
      out[i] = total;
    }
  }
}

// ---- compact spectral arrays <-> full half-spectrum planes -------------------------------------------------
// decomposition.py:233-236 stores level k as field[weights_k > 1e-12] (row-major).  mask_row_counts + one scan
// give the position of every row's first kept coefficient; expand_compact walks a row with ballots.
__global__ __launch_bounds__(kThreads) void mask_row_counts(const double *__restrict__ w, int nc, int *__restrict__ counts) {
  __shared__ int s_part[kThreads / 64];
  const double *row = w + static_cast<size_t>(blockIdx.x) * nc;
  int mine = 0;
  for (int c0 = 0; c0 < nc; c0 += kThreads) {
    const int c = c0 + threadIdx.x;
    mine += __popcll(__ballot(c < nc && row[c] > 1e-12));
  }
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int q = 0; q < kThreads / 64; ++q) t += s_part[q];
    counts[blockIdx.x] = t;
  }
}
// exclusive scan of m counts in place, total behind them (one workgroup; m <= 8192)
__global__ __launch_bounds__(kThreads) void scan_rows(int *__restrict__ counts, int m) {
  __shared__ int s_carry, s_wave[kThreads / 64];
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int i0 = 0; i0 < m; i0 += kThreads) {
    const int i = i0 + threadIdx.x;
    const int v = i < m ? counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if ((threadIdx.x & 63) >= d) incl += up;
    }
    if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
    __syncthreads();
    int before = s_carry;
    for (int q = 0; q < static_cast<int>(threadIdx.x >> 6); ++q) before += s_wave[q];
    if (i < m) counts[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == kThreads - 1) s_carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[m] = s_carry;
}
__global__ __launch_bounds__(kThreads) void expand_compact(const double *__restrict__ w, int nc, const int *__restrict__ offsets,
                                                           const double2 *__restrict__ src, double2 *__restrict__ dst) {
  __shared__ int s_part[kThreads / 64];
  const int r = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const double *row = w + static_cast<size_t>(r) * nc;
  double2 *out = dst + static_cast<size_t>(r) * nc;
  int base = offsets[r];
  for (int c0 = 0; c0 < nc; c0 += kThreads) {
    const int c = c0 + threadIdx.x;
    const bool kept = c < nc && row[c] > 1e-12;
    const unsigned long long mask = __ballot(kept);
    if (lane == 0) s_part[wave] = __popcll(mask);
    __syncthreads();
    int at = base;
    for (int q = 0; q < wave; ++q) at += s_part[q];
    if (c < nc) out[c] = kept ? src[at + __popcll(mask & ((1ull << lane) - 1ull))] : make_double2(0.0, 0.0);
    for (int q = 0; q < kThreads / 64; ++q) base += s_part[q];
    __syncthreads();
  }
}

__global__ __launch_bounds__(kThreads) void field_min_key(const double *__restrict__ field, size_t n,
                                                          unsigned long long *__restrict__ min_out) {
  const size_t stride = static_cast<size_t>(gridDim.x) * kThreads;
  unsigned long long key = ~0ull;
  for (size_t i = static_cast<size_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += stride) {
    const unsigned long long k = min_key(field[i]);
    key = k < key ? k : key;
  }
  publish_min(key, min_out);
}

}  // namespace
}  // namespace psh

using psh::fail;

static int ar_recompose_run(double *cascades_dev, int nlevels, int p, size_t plane, int head, const double *phi_host,
                            const double *eps_dev, const double *eps_stats_dev, const double *eps_scale_host,
                            const double *mu_host, const double *sigma_host, double *field_dev,
                            unsigned long long *min_key_dev) {
  PSH_REQUIRE_INIT();
  if (!cascades_dev || !phi_host || !mu_host || !sigma_host || !field_dev)
    return fail(PSH_EINVAL, "steps_ar_recompose: NULL pointer");
  if (nlevels < 1 || nlevels > psh::kMaxLevels || p < 1 || p > psh::kMaxOrder)
    return fail(PSH_EUNSUPPORTED, "steps_ar_recompose: 1..%d cascade levels, AR order 1..%d", psh::kMaxLevels, psh::kMaxOrder);
  if (plane == 0 || head < 0 || head >= p) return fail(PSH_EINVAL, "steps_ar_recompose: empty plane or ring head outside 0..p-1");
  if (eps_dev && !eps_scale_host) return fail(PSH_EINVAL, "steps_ar_recompose: eps without its scale factors");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::ArRecompose a;
  for (int k = 0; k < nlevels; ++k) {
    for (int j = 0; j <= p; ++j) a.phi[k][j] = phi_host[static_cast<size_t>(k) * (p + 1) + j];
    a.eps_scale[k] = eps_dev ? eps_scale_host[k] : 1.0;
    a.mu[k] = mu_host[k];
    a.sigma[k] = sigma_host[k];
  }
  a.nlevels = nlevels;
  a.p = p;
  a.head = head;
  a.has_eps = eps_dev != nullptr;
  if (min_key_dev) PSH_HIP(hipMemsetAsync(min_key_dev, 0xff, sizeof(unsigned long long), c.stream));
  hipLaunchKernelGGL(psh::ar_recompose, dim3(psh::grid_for(plane / 2 + 1)), dim3(psh::kThreads), 0, c.stream, cascades_dev, eps_dev,
                     reinterpret_cast<const double2 *>(eps_stats_dev), plane, a, field_dev, min_key_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_steps_ar_recompose_dev(double *cascades_dev, int nlevels, int p, size_t plane, int head,
                                          const double *phi_host, const double *eps_dev, const double *eps_scale_host,
                                          const double *mu_host, const double *sigma_host, double *field_dev,
                                          unsigned long long *min_key_dev) {
  return ar_recompose_run(cascades_dev, nlevels, p, plane, head, phi_host, eps_dev, nullptr, eps_scale_host, mu_host, sigma_host,
                          field_dev, min_key_dev);
}

extern "C" int psh_steps_ar_recompose_raw_dev(double *cascades_dev, int nlevels, int p, size_t plane, int head,
                                              const double *phi_host, const double *eps_dev, const double *eps_stats_dev,
                                              const double *eps_scale_host, const double *mu_host, const double *sigma_host,
                                              double *field_dev, unsigned long long *min_key_dev) {
  if (!eps_dev || !eps_stats_dev) return fail(PSH_EINVAL, "steps_ar_recompose_raw: the noise levels and their statistics are required");
  return ar_recompose_run(cascades_dev, nlevels, p, plane, head, phi_host, eps_dev, eps_stats_dev, eps_scale_host, mu_host,
                          sigma_host, field_dev, min_key_dev);
}

extern "C" int psh_steps_mask_dev(double *field_dev, size_t n, const double *grey_mask_dev, const unsigned char *keep_mask_dev,
                                  const unsigned long long *min_key_dev) {
  PSH_REQUIRE_INIT();
  if (!field_dev || !min_key_dev || (!grey_mask_dev == !keep_mask_dev))
    return fail(PSH_EINVAL, "steps_mask: field, minimum and exactly one of the two masks are required");
  if (n == 0) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::apply_mask, dim3(psh::grid_for(n)), dim3(psh::kThreads), 0, c.stream, field_dev, n, grey_mask_dev,
                     keep_mask_dev, min_key_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_steps_mean_shift_dev(double *field_dev, size_t n, double threshold, double mu_0) {
  PSH_REQUIRE_INIT();
  if (!field_dev) return fail(PSH_EINVAL, "steps_mean_shift: NULL pointer");
  if (n == 0) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const int nparts = 1024;
  void *partial = nullptr;
  if (int rc = psh_malloc(&partial, nparts * sizeof(double2))) return rc;
  hipLaunchKernelGGL(psh::wet_sum, dim3(nparts), dim3(psh::kThreads), 0, c.stream, field_dev, n, threshold,
                     static_cast<double2 *>(partial));
  hipLaunchKernelGGL(psh::mean_shift, dim3(psh::grid_for(n)), dim3(psh::kThreads), 0, c.stream, field_dev, n, threshold, mu_0,
                     static_cast<const double2 *>(partial), nparts);
  const hipError_t e = hipGetLastError();
  (void)psh_free(partial);  // stream-ordered
  PSH_HIP(e);
  return PSH_OK;
}

extern "C" int psh_ge_mask_dev(const double *field_dev, size_t n, double threshold, unsigned char *out_dev) {
  PSH_REQUIRE_INIT();
  if (!field_dev || !out_dev) return fail(PSH_EINVAL, "ge_mask: NULL pointer");
  if (n == 0) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::ge_mask, dim3(psh::grid_for(n)), dim3(psh::kThreads), 0, c.stream, field_dev, n, threshold, out_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_nan_where_dev(double *field_dev, const unsigned char *mask_dev, size_t n) {
  PSH_REQUIRE_INIT();
  if (!field_dev || !mask_dev) return fail(PSH_EINVAL, "nan_where: NULL pointer");
  if (n == 0) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::nan_where, dim3(psh::grid_for(n)), dim3(psh::kThreads), 0, c.stream, field_dev, mask_dev, n);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_lerp_dev(const double *a_dev, const double *b_dev, double w, double *out_dev, size_t n) {
  PSH_REQUIRE_INIT();
  if (!a_dev || !b_dev || !out_dev) return fail(PSH_EINVAL, "lerp: NULL pointer");
  if (n == 0) return PSH_OK;
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::lerp, dim3(psh::grid_for(n)), dim3(psh::kThreads), 0, c.stream, a_dev, b_dev, w, out_dev, n);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

// ---- spectral member update (see the kernels above) -------------------------------------------------
extern "C" int psh_steps_spectral_sums_dev(const void *noise_spec_dev, const double *filter_dev, const double *weights_dev,
                                           int nlevels, int m, int n, double *sums_dev) {
  PSH_REQUIRE_INIT();
  if (!noise_spec_dev || !filter_dev || !weights_dev || !sums_dev) return fail(PSH_EINVAL, "steps_spectral_sums: NULL pointer");
  if (nlevels < 1 || nlevels > psh::kMaxLevels || m <= 0 || n <= 1) return fail(PSH_EUNSUPPORTED, "steps_spectral_sums: 1..%d levels", psh::kMaxLevels);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const int nc = n / 2 + 1;
  const int grid = std::min(m, 1024);  // rows are dealt to the workgroups
  void *partial = nullptr;
  if (int rc = psh_malloc(&partial, static_cast<size_t>(grid) * psh::kMaxLevels * sizeof(double))) return rc;
  hipLaunchKernelGGL(psh::spectral_level_sums, dim3(grid), dim3(psh::kThreads), 0, c.stream, static_cast<const double2 *>(noise_spec_dev),
                     filter_dev, weights_dev, nlevels, m, nc, (n & 1) == 0 ? 1 : 0, static_cast<double *>(partial));
  hipLaunchKernelGGL(psh::spectral_level_sums_final, dim3(psh::kMaxLevels), dim3(psh::kThreads), 0, c.stream,
                     static_cast<const double *>(partial), grid, sums_dev);
  const hipError_t e = hipGetLastError();
  (void)psh_free(partial);  // stream-ordered
  PSH_HIP(e);
  return PSH_OK;
}

extern "C" int psh_steps_spectral_ar_dev(void *cascades_dev, int nlevels, int p, int m, int n, int head, const double *phi_host,
                                         const void *noise_spec_dev, const double *filter_dev, const double *weights_dev,
                                         const double *sums_dev, const double *noise_std_host, const double *mu_host,
                                         const double *sigma_host, void *field_spec_dev) {
  PSH_REQUIRE_INIT();
  if (!cascades_dev || !phi_host || !noise_spec_dev || !filter_dev || !weights_dev || !sums_dev || !noise_std_host || !mu_host ||
      !sigma_host || !field_spec_dev)
    return fail(PSH_EINVAL, "steps_spectral_ar: NULL pointer");
  if (nlevels < 1 || nlevels > psh::kMaxLevels || p < 1 || p > psh::kMaxOrder)
    return fail(PSH_EUNSUPPORTED, "steps_spectral_ar: 1..%d cascade levels, AR order 1..%d", psh::kMaxLevels, psh::kMaxOrder);
  if (m <= 0 || n <= 1 || head < 0 || head >= p) return fail(PSH_EINVAL, "steps_spectral_ar: invalid shape or ring head");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * (n / 2 + 1);
  const double mn = static_cast<double>(m) * static_cast<double>(n);
  psh::SpectralAr a;
  double mu_sum = 0.0;
  for (int k = 0; k < nlevels; ++k) {
    for (int j = 0; j <= p; ++j) a.phi[k][j] = phi_host[static_cast<size_t>(k) * (p + 1) + j];
    a.gain[k] = a.phi[k][p] * noise_std_host[k] * mn;
    a.sigma[k] = sigma_host[k];
    mu_sum += mu_host[k];
  }
  a.dc_add = mu_sum * mn;
  a.nlevels = nlevels;
  a.p = p;
  a.head = head;
  hipLaunchKernelGGL(psh::spectral_ar, dim3(psh::grid_for(plane)), dim3(psh::kThreads), 0, c.stream, static_cast<double2 *>(cascades_dev),
                     static_cast<const double2 *>(noise_spec_dev), filter_dev, weights_dev, sums_dev, plane, a,
                     static_cast<double2 *>(field_spec_dev));
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_field_min_key_dev(const double *field_dev, size_t n, unsigned long long *min_key_dev) {
  PSH_REQUIRE_INIT();
  if (!field_dev || !min_key_dev || n == 0) return fail(PSH_EINVAL, "field_min_key: NULL pointer or empty field");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipMemsetAsync(min_key_dev, 0xff, sizeof(unsigned long long), c.stream));
  // (few workgroups: every wave ends with an atomic on one address)
  hipLaunchKernelGGL(psh::field_min_key, dim3(std::min<unsigned>(psh::grid_for(n), 1024u)), dim3(psh::kThreads), 0, c.stream, field_dev, n,
                     min_key_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_steps_phase_ar_dev(void *cascades_dev, int nlevels, int p, int m, int n, int head, const double *phi_host,
                                      const double *theta_dev, const double *filter_dev, const double *weights_dev,
                                      double inv_std_noise, const double *inv_std_levels_host, const double *noise_std_host,
                                      const double *mu_host, const double *sigma_host, void *field_spec_dev) {
  PSH_REQUIRE_INIT();
  if (!cascades_dev || !phi_host || !weights_dev || !mu_host || !sigma_host || !field_spec_dev)
    return fail(PSH_EINVAL, "steps_phase_ar: NULL pointer");
  if (theta_dev && (!filter_dev || !inv_std_levels_host || !noise_std_host))
    return fail(PSH_EINVAL, "steps_phase_ar: phases without the noise filter and its constants");
  if (nlevels < 1 || nlevels > psh::kMaxLevels || p < 1 || p > psh::kMaxOrder)
    return fail(PSH_EUNSUPPORTED, "steps_phase_ar: 1..%d cascade levels, AR order 1..%d", psh::kMaxLevels, psh::kMaxOrder);
  if (m <= 0 || n <= 1 || head < 0 || head >= p) return fail(PSH_EINVAL, "steps_phase_ar: invalid shape or ring head");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::PhaseAr a;
  for (int k = 0; k < psh::kMaxLevels; ++k) {
    for (int j = 0; j <= psh::kMaxOrder; ++j) a.phi[k][j] = (k < nlevels && j <= p) ? phi_host[static_cast<size_t>(k) * (p + 1) + j] : 0.0;
    a.inv_std[k] = (theta_dev && k < nlevels) ? inv_std_levels_host[k] : 0.0;
    a.noise_std[k] = (theta_dev && k < nlevels) ? noise_std_host[k] : 0.0;
    a.mu[k] = k < nlevels ? mu_host[k] : 0.0;
    a.sigma[k] = k < nlevels ? sigma_host[k] : 0.0;
  }
  a.inv_stdn = inv_std_noise;
  a.nlevels = nlevels;
  a.p = p;
  a.head = head;
  a.m = m;
  a.nc = n / 2 + 1;
  hipLaunchKernelGGL(psh::spectral_phase_ar, dim3(std::min(m, 4096)), dim3(psh::kThreads), 0, c.stream, static_cast<double2 *>(cascades_dev),
                     theta_dev, filter_dev, weights_dev, a, static_cast<double2 *>(field_spec_dev));
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

// Level planes from the reference's compact spectral arrays (decomposition.py:233-236: field[weights > 1e-12]).
//   psh_mask_row_offsets_dev     offsets (m + 1) int32: position of every row's first kept coefficient in the compact
//                                array of this level, the number of kept coefficients behind them
//   psh_expand_compact_c128_dev  dst (m, nc) complex128 = src scattered to the kept coefficients, zero elsewhere
extern "C" int psh_mask_row_offsets_dev(const double *weights_dev, int m, int nc, int *offsets_dev) {
  PSH_REQUIRE_INIT();
  if (!weights_dev || !offsets_dev) return fail(PSH_EINVAL, "mask_row_offsets: NULL pointer");
  if (m <= 0 || m > 8192 || nc <= 0) return fail(PSH_EUNSUPPORTED, "mask_row_offsets: 1..8192 rows");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::mask_row_counts, dim3(m), dim3(psh::kThreads), 0, c.stream, weights_dev, nc, offsets_dev);
  hipLaunchKernelGGL(psh::scan_rows, dim3(1), dim3(psh::kThreads), 0, c.stream, offsets_dev, m);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_expand_compact_c128_dev(const double *weights_dev, int m, int nc, const int *offsets_dev, const void *src_dev,
                                           void *dst_dev) {
  PSH_REQUIRE_INIT();
  if (!weights_dev || !offsets_dev || !src_dev || !dst_dev) return fail(PSH_EINVAL, "expand_compact: NULL pointer");
  if (m <= 0 || nc <= 0) return fail(PSH_EINVAL, "expand_compact: invalid shape");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::expand_compact, dim3(m), dim3(psh::kThreads), 0, c.stream, weights_dev, nc, offsets_dev,
                     static_cast<const double2 *>(src_dev), static_cast<double2 *>(dst_dev));
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}
