// Whole dense Lucas-Kanade estimate in one C-ABI call.
//
// Orchestration of pysteps/motion/lucaskanade.py:182-279 (dense_lucaskanade) for the default
// detector / interpolator pair, built from the stage entry points of this library
// (psh_lk_*, psh_outliers_local_host, psh_decluster_host, psh_idw_dev).  It exists to keep the
// interpreter out of the critical path: between the four device->host hand-offs of the sparse
// stage only a few microseconds of C++ run instead of ~0.3 ms of Python and ctypes marshalling
// per estimate.  The Python shim (pysteps_amd/motion/lucaskanade.py) remains the reference
// mirror and falls back to its own stage-by-stage loop for anything this call does not take.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

namespace {

// RAII for psh_malloc blocks (stream-ordered free)
struct DevBlock {
  void *p = nullptr;
  ~DevBlock() {
    if (p) (void)psh_free(p);
  }
  int alloc(size_t bytes) { return psh_malloc(&p, bytes); }
  template <class T>
  T *as() const {
    return static_cast<T *>(p);
  }
};

__global__ __launch_bounds__(256) void fill_two_planes(float *out, size_t plane, float a, float b) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < 2 * plane; i += stride)
    out[i] = i < plane ? a : b;
}

int fill_field(float *out_dev, size_t plane, float a, float b) {
  psh::Context &c = psh::ctx();
  hipLaunchKernelGGL(fill_two_planes, dim3(2048), dim3(256), 0, c.stream, out_dev, plane, a, b);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

}  // namespace

extern "C" int psh_dense_lk_dev(const float *frames_dev, int nframes, int m, int n,
                                const psh_lk_params *prm, float *field_dev, double *xy_host,
                                double *uv_host, int capacity, int *count_out) {
  PSH_REQUIRE_INIT();
  if (!frames_dev || !prm) return psh::fail(PSH_EINVAL, "dense_lk: NULL pointer");
  if (nframes < 1 || m <= 0 || n <= 0) return psh::fail(PSH_EINVAL, "dense_lk: invalid shape");
  if (!field_dev && !(xy_host && uv_host && count_out))
    return psh::fail(PSH_EINVAL, "dense_lk: neither a dense output nor sparse output buffers given");
  if (prm->max_corners <= 0) return psh::fail(PSH_EINVAL, "dense_lk: max_corners must be positive");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const size_t plane = static_cast<size_t>(m) * n;
  // PYSTEPS_HIP_TRACE=1: host-side timeline of the call on stderr (where the host waits and works)
  static const bool trace = std::getenv("PYSTEPS_HIP_TRACE") != nullptr;
  const auto t_start = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (trace)
      std::fprintf(stderr, "dense_lk %-28s +%8.1f us\n", what,
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count());
  };

  // ---- per frame: cleaning + uint8 renderings (lucaskanade.py:213-224) -----------------
  std::vector<DevBlock> clean(nframes), trk(nframes), feat(nframes), stats(nframes);
  for (int t = 0; t < nframes; ++t) {
    const bool want_feat = t < nframes - 1;
    if (int rc = clean[t].alloc(plane * sizeof(float))) return rc;
    if (int rc = trk[t].alloc(plane)) return rc;
    if (want_feat)
      if (int rc = feat[t].alloc(plane)) return rc;
    if (int rc = stats[t].alloc(8 * sizeof(float))) return rc;
    if (int rc = psh_lk_prepare_dev(frames_dev + static_cast<size_t>(t) * plane, m, n, prm->size_opening,
                                    prm->buffer_mask, clean[t].as<float>(), trk[t].as<unsigned char>(),
                                    want_feat ? feat[t].as<unsigned char>() : nullptr, stats[t].as<float>()))
      return rc;
  }

  // ---- per frame pair: features, tracking, pooling (:207-242) ---------------------------
  // The successful tracks of all pairs are pooled ON THE DEVICE (lk_pool_append) and the outlier
  // test reads its sample count from device memory, so the only host hand-offs left are the
  // ordered corner pass of each pair and ONE copy of (count, xy, uv, flags) at the end.
  const int pairs = nframes - 1;
  const int capacity_dev = prm->max_corners * (pairs > 0 ? pairs : 1);
  if (capacity_dev > 8192)
    return psh::fail(PSH_EUNSUPPORTED, "dense_lk: more than 8192 pooled vectors (max_corners x frame pairs)");
  const size_t cap = static_cast<size_t>(capacity_dev);
  const size_t off_uv = cap * 16, off_cnt = 2 * cap * 16, off_fl = off_cnt + 256;
  DevBlock pool;
  if (int rc = pool.alloc(off_fl + cap)) return rc;
  char *pbase = pool.as<char>();
  double *d_pxy = reinterpret_cast<double *>(pbase), *d_puv = reinterpret_cast<double *>(pbase + off_uv);
  int *d_pcnt = reinterpret_cast<int *>(pbase + off_cnt);
  unsigned char *d_pfl = reinterpret_cast<unsigned char *>(pbase + off_fl);
  PSH_HIP(hipMemsetAsync(d_pcnt, 0, sizeof(int), c.stream));
  std::vector<float> pts(static_cast<size_t>(prm->max_corners) * 2);
  // Frame pairs in groups: the corner requests and pyramids of a whole group are queued first,
  // then each pair's ordered host pass overlaps the device work of the pairs behind it.
  const int group = psh::lk_corners_in_flight_limit();
  psh::lk_corners_drain();  // nothing stale from an earlier failure
  for (int t0 = 0; t0 + 1 < nframes; t0 += group) {
    const int t1 = std::min(nframes - 1, t0 + group);
    std::vector<void *> pyrs(static_cast<size_t>(t1 - t0), nullptr);
    auto fail_out = [&](int code) {
      psh::lk_corners_drain();
      for (void *h : pyrs) (void)psh_lk_pyramids_free(h);
      return code;
    };
    for (int t = t0; t < t1; ++t) {
      if (int rc = psh_lk_corners_launch_dev(feat[t].as<unsigned char>(), clean[t].as<float>(),
                                             stats[t].as<float>(), m, n, prm->block_size, prm->buffer_mask,
                                             prm->quality_level, prm->min_distance, prm->max_corners))
        return fail_out(rc);
    }
    for (int t = t0; t < t1; ++t) {  // built on the device while the host orders the corner candidates
      if (int rc = psh_lk_pyramids_dev(trk[t].as<unsigned char>(), trk[t + 1].as<unsigned char>(), m, n,
                                       prm->win_w, prm->win_h, prm->max_level, &pyrs[t - t0]))
        return fail_out(rc);
    }
    mark("pairs queued");
    for (int t = t0; t < t1; ++t) {
      int npts = 0;
      if (int rc = psh_lk_corners_finish(pts.data(), &npts)) return fail_out(rc);
      mark("corners ordered");
      if (npts > 0) {
        if (int rc = psh::lk_track_pool(pyrs[t - t0], pts.data(), npts, prm->max_count, prm->epsilon,
                                        prm->min_eig_threshold, d_pxy, d_puv, d_pcnt, capacity_dev))
          return fail_out(rc);
      }
      const int rc3 = psh_lk_pyramids_free(pyrs[t - t0]);  // stream-ordered: the tracker above is queued first
      pyrs[t - t0] = nullptr;
      if (rc3) return fail_out(rc3);
    }
  }
  // ---- outlier removal (:252-254) on the pooled vectors, then one hand-off to the host ------
  PSH_HIP(psh::launch_outliers_pooled(d_pxy, d_puv, d_pcnt, capacity_dev, prm->k_outlier, prm->nr_std_outlier,
                                      d_pfl, c.stream));
  // one copy of the whole pool block (xy | uv | count | flags), same layout on both sides
  static void *pinned = nullptr;  // sized for 8192 vectors
  constexpr size_t kPinBytes = 2 * 8192 * 16 + 256 + 8192;
  if (!pinned) PSH_HIP(hipHostMalloc(&pinned, kPinBytes, hipHostMallocDefault));
  char *pin = static_cast<char *>(pinned);
  PSH_HIP(hipMemcpyAsync(pin, pbase, off_fl + cap, hipMemcpyDeviceToHost, c.stream));
  mark("tracking queued");
  PSH_HIP(hipStreamSynchronize(c.stream));
  mark("pooled vectors on the host");
  const int pooled = std::min(*reinterpret_cast<const int *>(pin + off_cnt), capacity_dev);
  const double *hxy = reinterpret_cast<const double *>(pin);
  const double *huv = reinterpret_cast<const double *>(pin + off_uv);
  const unsigned char *hfl = reinterpret_cast<const unsigned char *>(pin + off_fl);
  std::vector<double> xy, uv;
  xy.reserve(2 * static_cast<size_t>(pooled));
  uv.reserve(2 * static_cast<size_t>(pooled));
  for (int i = 0; i < pooled; ++i) {
    if (pooled >= 2 && hfl[i]) continue;  // fewer than two samples: nothing is an outlier (:178-179)
    xy.push_back(hxy[2 * i]);
    xy.push_back(hxy[2 * i + 1]);
    uv.push_back(huv[2 * i]);
    uv.push_back(huv[2 * i + 1]);
  }
  int count = static_cast<int>(xy.size() / 2);
  if (!field_dev) {  // sparse vectors requested (dense=False, :260-261)
    if (count > capacity) return psh::fail(PSH_EINVAL, "dense_lk: %d vectors exceed the output capacity %d", count, capacity);
    for (int i = 0; i < 2 * count; ++i) {
      xy_host[i] = xy[i];
      uv_host[i] = uv[i];
    }
    *count_out = count;
    return PSH_OK;
  }

  // ---- declustering (:264-265) and interpolation (:272-274) ----------------------------------
  if (count > 0 && prm->decl_scale > 1.0) {
    std::vector<double> dxy(static_cast<size_t>(count) * 2), duv(static_cast<size_t>(count) * 2);
    int kept = 0;
    if (int rc = psh_decluster_host(xy.data(), uv.data(), count, prm->decl_scale, 1, dxy.data(), duv.data(), &kept))
      return rc;
    xy.assign(dxy.begin(), dxy.begin() + 2 * kept);
    uv.assign(duv.begin(), duv.begin() + 2 * kept);
    count = kept;
  }
  if (count_out) *count_out = count;
  if (count == 0) return fill_field(field_dev, plane, 0.f, 0.f);  // :245-249, :268-269
  double vmin = uv[0], vmax = uv[0];
  for (int i = 1; i < 2 * count; ++i) {
    vmin = std::fmin(vmin, uv[i]);
    vmax = std::fmax(vmax, uv[i]);
  }
  if (count == 1) return fill_field(field_dev, plane, static_cast<float>(uv[0]), static_cast<float>(uv[1]));
  if (vmin == vmax)  // "all equal elements" of the interpolator preamble (decorators.py:207-208)
    return fill_field(field_dev, plane, static_cast<float>(uv[0]), static_cast<float>(uv[0]));

  std::vector<float> fxy(static_cast<size_t>(count) * 2), fuv(fxy.size());
  double xmin = 0.0, xmax = n - 1.0, ymin = 0.0, ymax = m - 1.0;
  for (int i = 0; i < count; ++i) {
    fxy[2 * i] = static_cast<float>(xy[2 * i]);
    fxy[2 * i + 1] = static_cast<float>(xy[2 * i + 1]);
    fuv[2 * i] = static_cast<float>(uv[2 * i]);
    fuv[2 * i + 1] = static_cast<float>(uv[2 * i + 1]);
    xmin = std::fmin(xmin, xy[2 * i]);
    xmax = std::fmax(xmax, xy[2 * i]);
    ymin = std::fmin(ymin, xy[2 * i + 1]);
    ymax = std::fmax(ymax, xy[2 * i + 1]);
  }
  DevBlock samples;
  const size_t sbytes = fxy.size() * sizeof(float);
  if (int rc = samples.alloc(2 * sbytes)) return rc;
  float *d_xy = samples.as<float>(), *d_uv = d_xy + fxy.size();
  // pinned staging ring: the uploads are asynchronous and outlive this call's vectors
  static void *up_ring = nullptr;
  static size_t up_slot = 0;
  constexpr size_t kUpSlots = 8, kUpSlotBytes = 2 * 8192 * 2 * sizeof(float);
  if (!up_ring) PSH_HIP(hipHostMalloc(&up_ring, kUpSlots * kUpSlotBytes, hipHostMallocDefault));
  if (up_slot == kUpSlots) {
    PSH_HIP(hipStreamSynchronize(c.stream));
    up_slot = 0;
  }
  char *up = static_cast<char *>(up_ring) + (up_slot++) * kUpSlotBytes;
  std::memcpy(up, fxy.data(), sbytes);
  std::memcpy(up + sbytes, fuv.data(), sbytes);
  PSH_HIP(hipMemcpyAsync(d_xy, up, 2 * sbytes, hipMemcpyHostToDevice, c.stream));
  const double reach = std::hypot(xmax - xmin, ymax - ymin) * 1.001 + 1.0;
  const int k = prm->idw_k <= 0 ? count : prm->idw_k;
  mark("vectors declustered, uploaded");
  return psh_idw_dev(d_xy, d_uv, count, m, n, 0.0, 1.0, 0.0, 1.0, k, prm->idw_power, prm->idw_dist_offset, reach,
                     field_dev);
}
