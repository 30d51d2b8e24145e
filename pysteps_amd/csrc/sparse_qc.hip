// Local multivariate outlier test of sparse motion vectors on the device.
//
// Replaces the form of pysteps/utils/cleansing.py:124-249 (detect_outliers) that
// dense_lucaskanade uses (pysteps/motion/lucaskanade.py:252: uv (N,2), coord xy
// (N,2), k nearest neighbours): for every vector the k nearest OTHER vectors
// (cKDTree k+1 query minus the first hit, :221-224), z = uv - mean(neighbours),
// C = cov(neighbours, ddof=1), outlier iff sqrt(z^T C^-1 z) > thr; singular C ->
// not an outlier (:239-243).
//
// N is a few thousand at most.  One WAVE owns one vector: the 64 lanes compute the
// N squared distances into an LDS key array (distance bits << 32 | index, so that a
// single 64-bit minimum also breaks ties by the lower index - cKDTree's own tie
// order is unspecified), then the k+1 nearest are extracted one by one with a wave
// minimum; the owner lane retires the key and rescans its N/64 entries.  Sums for
// mean / covariance are accumulated in float64 relative to the vector itself.
// Runtime ~ 10 us; it exists to keep the 2-4 ms host k-d tree query off the critical
// path of a nowcast step.
#include <vector>

#include "common.h"

namespace psh {
namespace {

constexpr int kQcMaxN = 8192;  // LDS keys: 8 B per vector (64 KiB)

// Wave-wide minimum of a 32-bit unsigned value in six DPP steps (pure VALU: a __shfl_xor goes
// through the LDS crossbar and costs ~100 clk each, twelve of them per 64-bit step): xor-1 / xor-2
// inside the quads, half-mirror and mirror inside each row of 16, row_bcast15 / row_bcast31 across
// the rows; lane 63 then holds the minimum, which is broadcast through an SGPR.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define PSH_MIN_STEP(CTRL, ROWMASK)                                                                       \
  {                                                                                                        \
    const unsigned o = static_cast<unsigned>(                                                              \
        __builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), CTRL, ROWMASK, 0xf, false)); \
    v = o < v ? o : v;                                                                                     \
  }
  PSH_MIN_STEP(0xB1, 0xf)   // quad_perm:[1,0,3,2]
  PSH_MIN_STEP(0x4E, 0xf)   // quad_perm:[2,3,0,1]
  PSH_MIN_STEP(0x141, 0xf)  // row_half_mirror
  PSH_MIN_STEP(0x140, 0xf)  // row_mirror
  PSH_MIN_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1 and 3
  PSH_MIN_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2 and 3
#undef PSH_MIN_STEP
  return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// lexicographic minimum of (distance bits << 32 | index) keys: the smallest distance first, then
// the smallest index among the lanes that hold it
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
  const unsigned hi = static_cast<unsigned>(v >> 32), lo = static_cast<unsigned>(v);
  const unsigned m_hi = wave_min_u32(hi);
  const unsigned m_lo = wave_min_u32(hi == m_hi ? lo : 0xffffffffu);
  return (static_cast<unsigned long long>(m_hi) << 32) | m_lo;
}

__global__ __launch_bounds__(64) void outliers_local(const double2 *__restrict__ xy,
                                                     const double2 *__restrict__ uv, int n_host,
                                                     const int *__restrict__ n_dev, int k,
                                                     double thr,
                                                     unsigned char *__restrict__ flags) {
  extern __shared__ unsigned long long keys[];
  // the sample count is either known on the host or still sitting in device memory (the
  // pooled tracker output); in the latter case the grid covers the capacity and extra
  // workgroups leave at once
  const int n = n_dev ? *n_dev : n_host;
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= n) return;
  if (n < 2) {
    if (lane == 0) flags[i] = 0;
    return;
  }
  const double2 me = xy[i], mine = uv[i];
  constexpr unsigned long long kGone = ~0ull;
  unsigned long long best = kGone;
  for (int j = lane; j < n; j += 64) {
    const double dx = xy[j].x - me.x, dy = xy[j].y - me.y;
    const float d = static_cast<float>(dx * dx + dy * dy);
    const unsigned long long key =
        (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | static_cast<unsigned>(j);
    keys[j] = key;
    best = key < best ? key : best;
  }
  double sa = 0.0, sb = 0.0, saa = 0.0, sab = 0.0, sbb = 0.0;
  int cnt = 0;
  const int kk = min(n, k + 1);  // nearest hits incl. the vector itself
  // The neighbours are extracted first (pure LDS / cross-lane work), their vectors are then
  // fetched by kk lanes at once - one round trip to memory instead of one per neighbour - and
  // summed by lane 0 in extraction order (the order the sums always had: bit-identical flags).
  __shared__ double2 s_nb[64];
  int mine_j = -1;
  for (int t0 = 0; t0 < kk; t0 += 64) {
    const int tn = min(64, kk - t0);
    for (int t = 0; t < tn; ++t) {
      const unsigned long long g = wave_min_u64(best);
      const int j = static_cast<int>(g & 0xffffffffull);
      if (lane == t) mine_j = j;
      if ((j & 63) == lane) {  // owner retires the key and rescans its entries
        keys[j] = kGone;
        best = kGone;
        // eight LDS reads in flight at a time: a single lane is active here, so the rescan is pure
        // latency (one read after the other cost 16 x ~100 clk per extraction at 1000 vectors)
        for (int q0 = lane; q0 < n; q0 += 8 * 64) {
          unsigned long long v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = q0 + u * 64 < n ? keys[q0 + u * 64] : kGone;
#pragma unroll
          for (int u = 0; u < 8; ++u) best = v[u] < best ? v[u] : best;
        }
      }
    }
    if (lane < tn) {
      const double2 q = uv[mine_j];
      s_nb[lane] = make_double2(q.x - mine.x, q.y - mine.y);
    }
    __syncthreads();
    if (lane == 0) {
      for (int t = (t0 == 0 ? 1 : 0); t < tn; ++t) {  // the first hit is the vector itself (or a duplicate position): dropped
        const double a = s_nb[t].x, b = s_nb[t].y;
        sa += a;
        sb += b;
        saa += a * a;
        sab += a * b;
        sbb += b * b;
        ++cnt;
      }
    }
    __syncthreads();
  }
  if (lane != 0) return;
  bool out = false;
  if (cnt >= 2) {
    const double ma = sa / cnt, mb = sb / cnt;  // neighbour mean relative to this vector
    const double dof = cnt - 1;
    const double caa = (saa - sa * ma) / dof, cab = (sab - sa * mb) / dof,
                 cbb = (sbb - sb * mb) / dof;
    const double det = caa * cbb - cab * cab;
    if (det != 0.0 && isfinite(det)) {
      const double zu = -ma, zv = -mb;  // this vector minus the neighbour mean
      const double md2 = (zu * zu * cbb - 2.0 * zu * zv * cab + zv * zv * caa) / det;
      out = sqrt(md2) > thr;
    }
  }
  flags[i] = out ? 1 : 0;
}

// The same test with every lane's keys SORTED once (register sorting network), for up to 64 x KPER
// vectors.  outliers_local above rescans the owner lane's entries after every extraction - a single
// active lane and two dependent LDS round trips, 31 times per vector; here the owner pops the next key
// of its sorted list (one LDS read).  Extraction order, sums and flags are those of outliers_local.
template <int N>
__device__ __forceinline__ void sort_keys_ascending(unsigned long long (&a)[N]) {
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const unsigned long long x = a[i], y = a[l];
          const bool swap = up ? (x > y) : (x < y);
          a[i] = swap ? y : x;
          a[l] = swap ? x : y;
        }
      }
    }
  }
}

template <int KPER>
__global__ __launch_bounds__(64) void outliers_local_sorted(const double2 *__restrict__ xy,
                                                            const double2 *__restrict__ uv, int n_host,
                                                            const int *__restrict__ n_dev, int k, double thr,
                                                            unsigned char *__restrict__ flags) {
  __shared__ unsigned long long keys[KPER * 64];  // keys[q * 64 + lane]: the lane's q-th smallest key
  __shared__ double2 s_nb[64];
  const int n = n_dev ? *n_dev : n_host;
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= n) return;
  if (n < 2) {
    if (lane == 0) flags[i] = 0;
    return;
  }
  const double2 me = xy[i], mine = uv[i];
  constexpr unsigned long long kGone = ~0ull;
  unsigned long long mykeys[KPER];
  double2 p[KPER];
#pragma unroll
  for (int q = 0; q < KPER; ++q) {  // (all loads in flight together)
    const int j = q * 64 + lane;
    p[q] = j < n ? xy[j] : make_double2(0.0, 0.0);
  }
#pragma unroll
  for (int q = 0; q < KPER; ++q) {
    const int j = q * 64 + lane;
    const double dx = p[q].x - me.x, dy = p[q].y - me.y;
    const float d = static_cast<float>(dx * dx + dy * dy);
    mykeys[q] = j < n ? (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | static_cast<unsigned>(j) : kGone;
  }
  sort_keys_ascending<KPER>(mykeys);
#pragma unroll
  for (int q = 0; q < KPER; ++q) keys[q * 64 + lane] = mykeys[q];
  unsigned long long best = mykeys[0];
  int next = 1;  // position of the lane's next key
  double sa = 0.0, sb = 0.0, saa = 0.0, sab = 0.0, sbb = 0.0;
  int cnt = 0;
  const int kk = min(n, k + 1);  // nearest hits incl. the vector itself
  int mine_j = -1;
  for (int t0 = 0; t0 < kk; t0 += 64) {
    const int tn = min(64, kk - t0);
    for (int t = 0; t < tn; ++t) {
      const unsigned long long g = wave_min_u64(best);
      const int j = static_cast<int>(g & 0xffffffffull);
      if (lane == t) mine_j = j;
      if ((j & 63) == lane) {  // the owner moves on to its next key
        best = next < KPER ? keys[next * 64 + lane] : kGone;
        ++next;
      }
    }
    if (lane < tn) {
      const double2 q = uv[mine_j];
      s_nb[lane] = make_double2(q.x - mine.x, q.y - mine.y);
    }
    __syncthreads();
    if (lane == 0) {
      for (int t = (t0 == 0 ? 1 : 0); t < tn; ++t) {  // the first hit is the vector itself (or a duplicate position): dropped
        const double a = s_nb[t].x, b = s_nb[t].y;
        sa += a;
        sb += b;
        saa += a * a;
        sab += a * b;
        sbb += b * b;
        ++cnt;
      }
    }
    __syncthreads();
  }
  if (lane != 0) return;
  bool out = false;
  if (cnt >= 2) {
    const double ma = sa / cnt, mb = sb / cnt;  // neighbour mean relative to this vector
    const double dof = cnt - 1;
    const double caa = (saa - sa * ma) / dof, cab = (sab - sa * mb) / dof,
                 cbb = (sbb - sb * mb) / dof;
    const double det = caa * cbb - cab * cab;
    if (det != 0.0 && isfinite(det)) {
      const double zu = -ma, zv = -mb;  // this vector minus the neighbour mean
      const double md2 = (zu * zu * cbb - 2.0 * zu * zv * cab + zv * zv * caa) / det;
      out = sqrt(md2) > thr;
    }
  }
  flags[i] = out ? 1 : 0;
}

// picks the kernel by the number of vectors the launch may see
static void launch_outliers(int rows, size_t cap_for_lds, hipStream_t stream, const double2 *xy, const double2 *uv, int n_host,
                            const int *n_dev, int k, double thr, unsigned char *flags) {
  if (rows <= 16 * 64) {
    hipLaunchKernelGGL(outliers_local_sorted<16>, dim3(rows), dim3(64), 0, stream, xy, uv, n_host, n_dev, k, thr, flags);
  } else if (rows <= 32 * 64) {
    hipLaunchKernelGGL(outliers_local_sorted<32>, dim3(rows), dim3(64), 0, stream, xy, uv, n_host, n_dev, k, thr, flags);
  } else {
    hipLaunchKernelGGL(outliers_local, dim3(rows), dim3(64), cap_for_lds * sizeof(unsigned long long), stream, xy, uv, n_host,
                       n_dev, k, thr, flags);
  }
}

}  // namespace
}  // namespace psh

namespace psh {
// device-resident form used by the one-call dense LK: count in device memory, capacity rows
hipError_t launch_outliers_pooled(const double *xy_dev, const double *uv_dev, const int *count_dev,
                                  int capacity, int k, double thr, unsigned char *flags_dev,
                                  hipStream_t stream) {
  launch_outliers(capacity, static_cast<size_t>(capacity), stream, reinterpret_cast<const double2 *>(xy_dev),
                  reinterpret_cast<const double2 *>(uv_dev), 0, count_dev, k, thr, flags_dev);
  return hipGetLastError();
}
}  // namespace psh

extern "C" int psh_outliers_local_host(const double *xy, const double *values, int n, int k,
                                       double thr, unsigned char *flags) {
  PSH_REQUIRE_INIT();
  if (n < 0) return psh::fail(PSH_EINVAL, "outliers: negative sample count");
  if (n == 0) return PSH_OK;
  if (!xy || !values || !flags) return psh::fail(PSH_EINVAL, "outliers: NULL pointer");
  if (k < 1) return psh::fail(PSH_EINVAL, "outliers: k must be >= 1");
  if (n > psh::kQcMaxN)
    return psh::fail(PSH_EUNSUPPORTED, "outliers: more than %d vectors not implemented on the device",
                     psh::kQcMaxN);
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
    psh::launch_outliers(n, static_cast<size_t>(n), c.stream, d_xy, d_uv, n, nullptr, k, thr, d_fl);
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipMemcpyAsync(flags, d_fl, static_cast<size_t>(n), hipMemcpyDeviceToHost, c.stream));
    PSH_HIP(hipStreamSynchronize(c.stream));
    return PSH_OK;
  };
  rc = run();
  (void)psh_free(blk);
  return rc;
}

// ---------------------------------------------------------------------------
// decluster (host): pysteps/utils/cleansing.py:21-121 for 2-d coordinates and
// 2-vectors with a scalar scale - cell = floor(xy / scale), cells in lexicographic
// (x_cell, y_cell) order (np.unique(axis=0), :101), per cell the component-wise
// median of the coordinates and of the vectors (:107-116).  A few thousand samples:
// plain C++ on the host (no device needed), ~20 us instead of ~0.4 ms of NumPy.
// ---------------------------------------------------------------------------
#include <algorithm>
#include <cmath>

namespace {
double median_of(std::vector<double> &v) {
  std::sort(v.begin(), v.end());
  const size_t c = v.size();
  return 0.5 * (v[(c - 1) / 2] + v[c / 2]);
}
}  // namespace

extern "C" int psh_decluster_host(const double *xy, const double *values, int n, double scale,
                                  int min_samples, double *out_xy, double *out_values,
                                  int *out_count) {
  if (n < 0) return psh::fail(PSH_EINVAL, "decluster: negative sample count");
  if (!out_count) return psh::fail(PSH_EINVAL, "decluster: NULL out_count");
  *out_count = 0;
  if (n == 0) return PSH_OK;
  if (!xy || !values || !out_xy || !out_values) return psh::fail(PSH_EINVAL, "decluster: NULL pointer");
  if (!(scale > 0.0) || !std::isfinite(scale)) return psh::fail(PSH_EINVAL, "decluster: scale must be positive");
  // Cells in lexicographic (x cell, y cell) order, samples of a cell in input order.
  static thread_local std::vector<long long> cx, cy;
  static thread_local std::vector<int> order;
  cx.resize(n);
  cy.resize(n);
  long long cx_lo = 0, cx_hi = 0, cy_lo = 0, cy_hi = 0;
  for (int i = 0; i < n; ++i) {
    const double fx = std::floor(xy[2 * i] / scale), fy = std::floor(xy[2 * i + 1] / scale);
    if (!(std::fabs(fx) < 4e15) || !(std::fabs(fy) < 4e15))
      return psh::fail(PSH_EINVAL, "decluster: coordinate / scale out of range");
    cx[i] = static_cast<long long>(fx);
    cy[i] = static_cast<long long>(fy);
    if (i == 0 || cx[i] < cx_lo) cx_lo = cx[i];
    if (i == 0 || cx[i] > cx_hi) cx_hi = cx[i];
    if (i == 0 || cy[i] < cy_lo) cy_lo = cy[i];
    if (i == 0 || cy[i] > cy_hi) cy_hi = cy[i];
  }
  order.resize(n);
  const unsigned long long nx = static_cast<unsigned long long>(cx_hi - cx_lo) + 1ull;
  const unsigned long long ny = static_cast<unsigned long long>(cy_hi - cy_lo) + 1ull;
  for (int i = 0; i < n; ++i) order[i] = i;
  if (nx <= 65536ull && ny <= 65536ull) {
    // stable LSD radix sort on the bytes of (y cell, then x cell) relative to the bounding box
    static thread_local std::vector<int> other;
    other.resize(n);
    int *src = order.data(), *dst = other.data();
    for (int pass = 0; pass < 4; ++pass) {
      const bool on_x = pass >= 2;
      const int shift = (pass & 1) * 8;
      if ((on_x ? nx : ny) <= (1ull << shift) && shift > 0) continue;  // the high byte is zero everywhere
      int count[257] = {0};
      auto digit = [&](int i) {
        return static_cast<int>(((on_x ? cx[i] - cx_lo : cy[i] - cy_lo) >> shift) & 0xff);
      };
      for (int t = 0; t < n; ++t) ++count[digit(src[t]) + 1];
      for (int d = 0; d < 256; ++d) count[d + 1] += count[d];
      for (int t = 0; t < n; ++t) dst[count[digit(src[t])]++] = src[t];
      std::swap(src, dst);
    }
    if (src != order.data()) std::copy(src, src + n, order.data());
  } else {
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      return cx[a] != cx[b] ? cx[a] < cx[b] : cy[a] < cy[b];
    });
  }
  std::vector<double> col;
  int written = 0;
  for (int s0 = 0; s0 < n;) {
    int e = s0 + 1;
    while (e < n && cx[order[e]] == cx[order[s0]] && cy[order[e]] == cy[order[s0]]) ++e;
    const int members = e - s0;
    if (members >= min_samples) {
      if (members == 1) {  // the median of one sample
        const int i = order[s0];
        out_xy[2 * written] = xy[2 * i];
        out_xy[2 * written + 1] = xy[2 * i + 1];
        out_values[2 * written] = values[2 * i];
        out_values[2 * written + 1] = values[2 * i + 1];
      } else {
        for (int c = 0; c < 4; ++c) {
          col.clear();
          for (int t = s0; t < e; ++t) {
            const int i = order[t];
            col.push_back(c < 2 ? xy[2 * i + c] : values[2 * i + (c - 2)]);
          }
          const double med = median_of(col);
          if (c < 2) {
            out_xy[2 * written + c] = med;
          } else {
            out_values[2 * written + (c - 2)] = med;
          }
        }
      }
      ++written;
    }
    s0 = e;
  }
  *out_count = written;
  return PSH_OK;
}
