// Whole dense Lucas-Kanade estimate in one C-ABI call.
//
// Orchestration of pysteps/motion/lucaskanade.py:182-279 (dense_lucaskanade) for the default
// detector / interpolator pair, built from the stage entry points of this library
// and the device-resident sparse stage (lk_sparse.hip).  The dense estimate is queued as ONE chain
// of kernel launches - no device->host copy, no host pass, no stream synchronisation in the
// middle - so the call returns while the GPU is still working and the motion field can feed the
// extrapolator directly.  The Python shim (pysteps_amd/motion/lucaskanade.py) remains the
// reference mirror and falls back to its own stage-by-stage loop for anything this call does not take.
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
__global__ __launch_bounds__(256) void zero_dwords(unsigned *__restrict__ p, size_t n) {
  for (size_t i = threadIdx.x; i < n; i += 256) p[i] = 0u;
}

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

}  // namespace

extern "C" int psh_dense_lk_uv_dev(const float *frames_dev, int nframes, int m, int n, const psh_lk_params *prm,
                                   float *field_dev, float *field_uv_dev, double *xy_host, double *uv_host, int capacity,
                                   int *count_out);

extern "C" int psh_dense_lk_dev(const float *frames_dev, int nframes, int m, int n,
                                const psh_lk_params *prm, float *field_dev, double *xy_host,
                                double *uv_host, int capacity, int *count_out) {
  return psh_dense_lk_uv_dev(frames_dev, nframes, m, n, prm, field_dev, nullptr, xy_host, uv_host, capacity, count_out);
}

// field_uv_dev (may be NULL; needs field_dev): the dense field once more as (m, n, 2) {u, v} pairs, written by the
// interpolation kernel itself - what psh_semilag_uv_dev takes instead of interleaving the planes on every call
extern "C" int psh_dense_lk_uv_dev(const float *frames_dev, int nframes, int m, int n, const psh_lk_params *prm,
                                   float *field_dev, float *field_uv_dev, double *xy_host, double *uv_host, int capacity,
                                   int *count_out) {
  PSH_REQUIRE_INIT();
  if (field_uv_dev && !field_dev) return psh::fail(PSH_EINVAL, "dense_lk: the interleaved field without the field itself");
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
  // Only the FIRST frame is prepared up front.  Frame t + 1 is cleaned and rendered on the side stream
  // beside the corner chain of frame t (which needs nothing of it): its passes stream the frame
  // through HBM while the Shi-Tomasi response is bound by the VALUs and the ordered walk is a single
  // workgroup; the pyramids of the pair follow on the same side stream, the tracker joins both.
  std::vector<DevBlock> clean(nframes), trk(nframes), feat(nframes), stats(nframes);
  const bool f64 = prm->frames_f64 != 0;
  const size_t frame_bytes = plane * (f64 ? sizeof(double) : sizeof(float));
  auto frame_ptr = [&](int t) { return reinterpret_cast<const char *>(frames_dev) + static_cast<size_t>(t) * frame_bytes; };
  for (int t = 0; t < nframes; ++t) {
    const bool want_feat = t < nframes - 1;
    // float32 frames: no cleaned copy in memory (one keep bit per pixel instead; the corner passes read the
    // NaN pattern from the frame itself).  float64 frames keep the float32 copy their passes write
    if (int rc = clean[t].alloc(f64 ? plane * sizeof(float) : psh::lk_keepbits_bytes(m, n))) return rc;
    if (int rc = trk[t].alloc(plane)) return rc;
    if (want_feat)
      if (int rc = feat[t].alloc(plane)) return rc;
    if (int rc = stats[t].alloc(8 * sizeof(float))) return rc;
  }
  const int pairs = nframes - 1;
  const int capacity_dev = prm->max_corners * (pairs > 0 ? pairs : 1);
  if (capacity_dev > 8192)
    return psh::fail(PSH_EUNSUPPORTED, "dense_lk: more than 8192 pooled vectors (max_corners x frame pairs)");
  if (!psh::corner_order_supported(m, n, prm->min_distance, prm->max_corners))
    return psh::fail(PSH_EUNSUPPORTED, "dense_lk: max_corners %d / image size beyond the resident corner pass",
                     prm->max_corners);
  if (field_dev && prm->idw_k > 32) return psh::fail(PSH_EUNSUPPORTED, "dense_lk: idw k=%d > 32", prm->idw_k);
  const size_t cap = static_cast<size_t>(capacity_dev);
  // pooled vectors (xy | uv | count | outlier flags: the block dense=False copies to the host as a whole),
  // behind them the statistic slots of every frame (common.h lk_slot_bytes).  Count and slots have to
  // start out as zero: ONE fill launch per call clears the tail of the block
  const size_t off_uv = cap * 16, off_cnt = 2 * cap * 16, off_fl = off_cnt + 256;
  const size_t slot_bytes = (psh::lk_slot_bytes() + 255) & ~static_cast<size_t>(255);
  const size_t off_slots = (off_fl + cap + 255) & ~static_cast<size_t>(255);
  DevBlock pool;
  if (int rc = pool.alloc(off_slots + slot_bytes * nframes)) return rc;
  char *pbase = pool.as<char>();
  double *d_pxy = reinterpret_cast<double *>(pbase), *d_puv = reinterpret_cast<double *>(pbase + off_uv);
  int *d_pcnt = reinterpret_cast<int *>(pbase + off_cnt);
  unsigned char *d_pfl = reinterpret_cast<unsigned char *>(pbase + off_fl);
  // (a kernel of its own instead of hipMemsetAsync: the runtime's fill comes with ~6 us of dispatch gap in front
  // of it, and it is the first thing on the estimate's critical path; all offsets are multiples of 256 bytes)
  hipLaunchKernelGGL(zero_dwords, dim3(1), dim3(256), 0, c.stream, reinterpret_cast<unsigned *>(pbase + off_cnt),
                     (off_slots + slot_bytes * nframes - off_cnt) / 4);
  PSH_HIP(hipGetLastError());
  auto frame_slots = [&](int t) { return reinterpret_cast<unsigned *>(pbase + off_slots + slot_bytes * t); };
  DevBlock ws_main, ws_side;  // allocated before any fork, released after the last join
  if (int rc = ws_main.alloc(psh::lk_prepare_ws_bytes(m, n, f64))) return rc;
  if (nframes > 1)
    if (int rc = ws_side.alloc(psh::lk_prepare_ws_bytes(m, n, f64))) return rc;
  auto prepare = [&](int t, hipStream_t stream, void *ws) {
    return psh::lk_prepare_on(stream, ws, frame_ptr(t), f64, m, n, prm->size_opening, prm->buffer_mask,
                              f64 ? clean[t].as<float>() : nullptr, trk[t].as<unsigned char>(),
                              t < nframes - 1 ? feat[t].as<unsigned char>() : nullptr, stats[t].as<float>(),
                              f64 ? nullptr : frame_slots(t), f64 ? nullptr : clean[t].as<unsigned long long>());
  };
  auto nan_pattern = [&](int t) { return f64 ? clean[t].as<float>() : reinterpret_cast<const float *>(frame_ptr(t)); };
  if (int rc = prepare(0, c.stream, ws_main.p)) return rc;

  // ---- per frame pair: features, tracking, pooling (:207-242) ---------------------------
  // Everything stays on the device: the corners are ordered and accepted by corner_order
  // (lk_sparse.hip), the tracker reads them and their count from device memory, the successful
  // tracks of all pairs are pooled by lk_pool_append and the outlier test reads its sample count
  // from device memory.  The dense estimate is one chain of kernel launches; only dense=False
  // (sparse vectors for the caller) ends with a copy to the host.
  DevBlock corners;  // [int count | pad to 256 | max_corners (x, y) float32], reused pair after pair
  if (int rc = corners.alloc(256 + static_cast<size_t>(prm->max_corners) * 2 * sizeof(float))) return rc;
  int *d_npts = corners.as<int>();
  float *d_pts = reinterpret_cast<float *>(corners.as<char>() + 256);
  for (int t = 0; t + 1 < nframes; ++t) {
    // side stream: the next frame's passes, then the pyramids of the pair (they only need the uint8
    // renderings); main stream: the corner chain of this frame; joined before the tracker.  The corner chain is
    // QUEUED first: its first kernel follows the frame passes on the main stream directly, and the ~10 launches of
    // the side stream cost the host more time than those passes run (round 4's timeline had the main stream idle
    // for 13 us in front of lk_corner_response_cols while the host was still queueing the side stream)
    hipStream_t side = nullptr;
    if (int rc = psh::side_begin(&side)) return rc;
    // (the pyramids' storage is taken NOW: a block the allocator hands out after the corner chain is queued may be
    // one that chain just gave back, and the side stream does not wait for the chain)
    const size_t pyr_bytes = psh::lk_pyramids_bytes(m, n, prm->win_w, prm->win_h, prm->max_level);
    void *pyr_block = nullptr;
    if (int rc = psh_malloc(&pyr_block, pyr_bytes)) {
      (void)psh::side_end();
      return rc;
    }
    int walk_stats[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int rc1 = psh::lk_corners_resident(feat[t].as<unsigned char>(), nan_pattern(t),
                                             stats[t].as<float>(), m, n, prm->block_size, prm->buffer_mask,
                                             prm->quality_level, prm->min_distance, prm->max_corners, d_pts, d_npts,
                                             nullptr, nullptr, trace ? walk_stats : nullptr, frame_slots(t));
    if (int rc = prepare(t + 1, side, ws_side.p)) {
      (void)psh::side_end();
      (void)psh_free(pyr_block);
      return rc;
    }
    struct Fork {
      void *pyr;
      int rc;
    } fork{nullptr, PSH_OK};
    // queued behind the frame passes on the side stream (forked above, after this frame's passes on the main stream)
    fork.rc = psh::lk_pyramids_on_side(side, pyr_block, pyr_bytes, trk[t].as<unsigned char>(), trk[t + 1].as<unsigned char>(),
                                       m, n, prm->win_w, prm->win_h, prm->max_level, &fork.pyr);
    const int rcj = psh::side_end();
    void *pyr = fork.pyr;
    if (trace)
      std::fprintf(stderr,
                   "dense_lk corner walk: %d chunk(s), %d candidates, %d evaluation(s) of the first wave; us: load %d sort %d "
                   "coordinates %d block tests %d survivors %d (cells %d - evaluations %d append %d) total %d\n",
                   walk_stats[0], walk_stats[1], walk_stats[2], walk_stats[3], walk_stats[4], walk_stats[5],
                   walk_stats[6], walk_stats[7], walk_stats[9], walk_stats[11], walk_stats[12], walk_stats[8]);
    if (rc1 || rcj || fork.rc || !pyr) {
      if (pyr) (void)psh_lk_pyramids_free(pyr);
      return fork.rc ? fork.rc : rc1 ? rc1 : rcj ? rcj : psh::fail(PSH_EHIP, "dense_lk: pyramids were not built");
    }
    const int rc2 = psh::lk_track_pool(pyr, nullptr, d_pts, d_npts, prm->max_corners, prm->max_count, prm->epsilon,
                                       prm->min_eig_threshold, d_pxy, d_puv, d_pcnt, capacity_dev);
    const int rc3 = psh_lk_pyramids_free(pyr);  // stream-ordered: the tracker above is queued first
    if (rc2) return rc2;
    if (rc3) return rc3;
  }
  mark("pairs queued");
  // ---- outlier removal (:252-254) on the pooled vectors --------------------------------------
  PSH_HIP(psh::launch_outliers_pooled(d_pxy, d_puv, d_pcnt, capacity_dev, prm->k_outlier, prm->nr_std_outlier,
                                      d_pfl, c.stream));
  if (field_dev) {
    // ---- dense field: filter, declustering (:264-265), interpolator preamble and IDW (:272-274),
    // all from device memory (vectors_finish writes the sample list and its length) -------------
    DevBlock samples;
    const size_t sbytes = cap * 2 * sizeof(float);
    if (int rc = samples.alloc(2 * sbytes + 256)) return rc;
    float *d_xy = samples.as<float>(), *d_uv = d_xy + cap * 2;
    psh::IdwDyn *d_dyn = reinterpret_cast<psh::IdwDyn *>(samples.as<char>() + 2 * sbytes);
    PSH_HIP(psh::launch_vectors_finish(d_pxy, d_puv, d_pfl, d_pcnt, capacity_dev, prm->decl_scale, m, n, d_xy, d_uv,
                                       d_dyn, c.stream));
    if (int rc = psh::idw_resident(d_xy, d_uv, capacity_dev, d_dyn, m, n, prm->idw_k, prm->idw_power,
                                   prm->idw_dist_offset, field_dev, field_uv_dev))
      return rc;
    mark("dense field queued");
    if (count_out) {  // the caller asks how many vectors the field was made from: one 4-byte copy
      static void *pinned_cnt = nullptr;
      if (int rc = psh::persistent_pinned(&pinned_cnt, 64)) return rc;
      PSH_HIP(hipMemcpyAsync(pinned_cnt, &d_dyn->L, sizeof(int), hipMemcpyDeviceToHost, c.stream));
      PSH_HIP(hipStreamSynchronize(c.stream));
      *count_out = *static_cast<const int *>(pinned_cnt);
    }
    return PSH_OK;
  }

  // ---- sparse vectors requested (dense=False, :260-261): one copy of the whole pool block
  // (xy | uv | count | flags), same layout on both sides ------------------------------------------
  static void *pinned = nullptr;  // sized for 8192 vectors
  constexpr size_t kPinBytes = 2 * 8192 * 16 + 256 + 8192;
  if (int rc = psh::persistent_pinned(&pinned, kPinBytes)) return rc;
  char *pin = static_cast<char *>(pinned);
  PSH_HIP(hipMemcpyAsync(pin, pbase, off_fl + cap, hipMemcpyDeviceToHost, c.stream));
  PSH_HIP(hipStreamSynchronize(c.stream));
  mark("pooled vectors on the host");
  const int pooled = std::min(*reinterpret_cast<const int *>(pin + off_cnt), capacity_dev);
  const double *hxy = reinterpret_cast<const double *>(pin);
  const double *huv = reinterpret_cast<const double *>(pin + off_uv);
  const unsigned char *hfl = reinterpret_cast<const unsigned char *>(pin + off_fl);
  int count = 0;
  for (int i = 0; i < pooled; ++i) {
    if (pooled >= 2 && hfl[i]) continue;  // fewer than two samples: nothing is an outlier (:178-179)
    if (count >= capacity)
      return psh::fail(PSH_EINVAL, "dense_lk: the vectors exceed the output capacity %d", capacity);
    xy_host[2 * count] = hxy[2 * i];
    xy_host[2 * count + 1] = hxy[2 * i + 1];
    uv_host[2 * count] = huv[2 * i];
    uv_host[2 * count + 1] = huv[2 * i + 1];
    ++count;
  }
  *count_out = count;
  return PSH_OK;
}
