// Host-buffer ("drop-in") path of the extrapolator: what pysteps' callers reach when they hand
// NumPy arrays to extrapolation.get_method("semilagrangian") (nowcasts/extrapolation.py:92,
// nowcasts/utils.py:453-458, steps.py:697-703).  The kernel needs 1.4 ms for a 4096^2 x 24
// nowcast, the 1.5 GiB of output planes need ~28 ms over PCIe Gen5: this path is transfer bound,
// so what matters here is that every byte crosses the bus at pinned-memory speed, once, and that
// the two directions of the link and the host's staging copies overlap.
//
//  * psh_host_alloc / psh_host_free: pinned host blocks from a cached pool.  The Python shim
//    allocates the RESULT arrays from it (NumPy arrays over pinned memory, returned to the pool by
//    a finaliser when the caller drops them), so the device-to-host copy lands directly in the
//    array the caller receives - no staging, no second pass over 1.5 GiB.
//  * Pointers that are not pinned (the caller's own input arrays; result buffers of foreign
//    callers of the C ABI) are staged through a ring of pinned chunks: several host threads copy
//    chunk c (one thread moves ~10 GB/s, the link ~55 GB/s) while the DMA engine moves chunk c-1.
//  * Downloads run on their own stream behind an event, uploads and the kernel on the library
//    stream; the context mutex is released while this thread waits or copies, so other caller
//    threads (dask workers, nowcasts/utils.py:464-468) can queue their work meanwhile.
//  * One kernel launch for all lead times: splitting it into lead-time groups so that the
//    download of group g overlaps the computation of group g+1 can hide at most the kernel's own
//    1.4 ms (4 % of the transfer time) and would cost a float64 displacement round trip per group.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <vector>

#include "common.h"

namespace psh {
namespace {

struct PinnedPool {
  std::mutex mu;
  std::map<size_t, std::vector<void *>> free_blocks;
  std::map<void *, size_t> live;
  size_t cached = 0, in_use = 0, limit = 0;
};

PinnedPool &pool() {
  static PinnedPool p;
  if (p.limit == 0) {
    const char *env = std::getenv("PYSTEPS_HIP_PINNED_BYTES");
    p.limit = env ? static_cast<size_t>(std::strtoull(env, nullptr, 10)) : (size_t(16) << 30);
    if (p.limit == 0) p.limit = 1;
  }
  return p;
}

constexpr size_t kChunk = size_t(32) << 20;  // staging chunk
constexpr int kSlots = 3;

// dst/src host memory, several threads: one core streams ~10 GB/s, the link wants ~55
void parallel_copy(void *dst, const void *src, size_t nbytes) {
  const size_t per = size_t(4) << 20;
  int nt = static_cast<int>(std::min<size_t>(8, (nbytes + per - 1) / per));
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw > 0 && hw < 16) nt = std::min(nt, std::max(1, static_cast<int>(hw) / 2));
  if (nt <= 1) {
    std::memcpy(dst, src, nbytes);
    return;
  }
  std::vector<std::thread> th;
  th.reserve(nt - 1);
  const size_t share = ((nbytes / nt) + 4095) & ~size_t(4095);
  for (int t = 1; t < nt; ++t) {
    const size_t off = share * t;
    if (off >= nbytes) break;
    const size_t len = std::min(share, nbytes - off);
    th.emplace_back([=] { std::memcpy(static_cast<char *>(dst) + off, static_cast<const char *>(src) + off, len); });
  }
  std::memcpy(dst, src, std::min(share, nbytes));
  for (auto &t : th) t.join();
}

bool is_pinned(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();  // plain pageable memory: not an error for us
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

hipStream_t g_down_stream = nullptr;  // device-to-host copies of the host path (created under the lock)

// ring of pinned staging chunks + one event per slot
struct Ring {
  void *block = nullptr;
  hipEvent_t ev[kSlots] = {};
  bool used[kSlots] = {};
  int next = 0;
  int init() {
    if (block) return PSH_OK;
    if (int rc = pinned_alloc(&block, kSlots * kChunk)) return rc;
    for (int i = 0; i < kSlots; ++i) PSH_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    return PSH_OK;
  }
  char *slot(int i) { return static_cast<char *>(block) + static_cast<size_t>(i) * kChunk; }
  void destroy() {
    for (int i = 0; i < kSlots; ++i)
      if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (block) (void)pinned_free(block);
    block = nullptr;
  }
};

using Lock = std::unique_lock<std::recursive_mutex>;

// host -> device on `stream`; pinned sources go straight, pageable ones through the ring
int upload(void *dst_dev, const void *src, size_t nbytes, hipStream_t stream, Ring &ring, Lock &lock) {
  if (nbytes == 0) return PSH_OK;
  if (is_pinned(src)) {
    PSH_HIP(hipMemcpyAsync(dst_dev, src, nbytes, hipMemcpyHostToDevice, stream));
    return PSH_OK;
  }
  if (int rc = ring.init()) return rc;
  for (size_t off = 0; off < nbytes; off += kChunk) {
    const size_t len = std::min(kChunk, nbytes - off);
    const int s = ring.next;
    ring.next = (ring.next + 1) % kSlots;
    lock.unlock();  // CPU work and waits: let other caller threads queue their calls
    if (ring.used[s]) (void)hipEventSynchronize(ring.ev[s]);
    parallel_copy(ring.slot(s), static_cast<const char *>(src) + off, len);
    lock.lock();
    PSH_HIP(hipMemcpyAsync(static_cast<char *>(dst_dev) + off, ring.slot(s), len, hipMemcpyHostToDevice, stream));
    PSH_HIP(hipEventRecord(ring.ev[s], stream));
    ring.used[s] = true;
  }
  return PSH_OK;
}

// device -> host on the download stream (which already waits for the producer); returns after the
// data is in `dst`
int download(void *dst, const void *src_dev, size_t nbytes, Ring &ring, Lock &lock) {
  if (nbytes == 0) return PSH_OK;
  if (is_pinned(dst)) {
    PSH_HIP(hipMemcpyAsync(dst, src_dev, nbytes, hipMemcpyDeviceToHost, g_down_stream));
    return PSH_OK;  // the caller synchronises the download stream once at the end
  }
  if (int rc = ring.init()) return rc;
  // slots of the ring may still feed uploads: those were queued on the library stream before the
  // kernel the download stream waits for, so they are complete when the first chunk lands
  const size_t nchunks = (nbytes + kChunk - 1) / kChunk;
  for (size_t c = 0; c <= nchunks; ++c) {
    if (c < nchunks) {
      const size_t off = c * kChunk, len = std::min(kChunk, nbytes - off);
      const int s = static_cast<int>(c % kSlots);
      PSH_HIP(hipMemcpyAsync(ring.slot(s), static_cast<const char *>(src_dev) + off, len, hipMemcpyDeviceToHost,
                             g_down_stream));
      PSH_HIP(hipEventRecord(ring.ev[s], g_down_stream));
      ring.used[s] = true;
    }
    if (c >= 1) {  // chunk c-1: wait for the DMA, then fan it out to the caller's pages
      const size_t off = (c - 1) * kChunk, len = std::min(kChunk, nbytes - off);
      const int s = static_cast<int>((c - 1) % kSlots);
      lock.unlock();
      const hipError_t e = hipEventSynchronize(ring.ev[s]);
      if (e == hipSuccess) parallel_copy(static_cast<char *>(dst) + off, ring.slot(s), len);
      lock.lock();
      PSH_HIP(e);
    }
  }
  return PSH_OK;
}

}  // namespace

int pinned_alloc(void **host_ptr, size_t nbytes) {
  *host_ptr = nullptr;
  if (nbytes == 0) return PSH_OK;
  nbytes = (nbytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);
  PinnedPool &p = pool();
  std::lock_guard<std::mutex> g(p.mu);
  auto it = p.free_blocks.find(nbytes);
  if (it != p.free_blocks.end() && !it->second.empty()) {
    *host_ptr = it->second.back();
    it->second.pop_back();
    p.cached -= nbytes;
  } else {
    if (p.in_use + p.cached + nbytes > p.limit) {  // make room from the cache first
      for (auto &kv : p.free_blocks)
        for (void *q : kv.second) (void)hipHostFree(q);
      p.free_blocks.clear();
      p.cached = 0;
    }
    if (p.in_use + nbytes > p.limit)
      return fail(PSH_ENOMEM, "pinned host pool: %zu bytes in use, limit %zu (PYSTEPS_HIP_PINNED_BYTES)", p.in_use,
                  p.limit);
    const hipError_t e = hipHostMalloc(host_ptr, nbytes, hipHostMallocDefault);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      *host_ptr = nullptr;
      return fail(PSH_ENOMEM, "hipHostMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
    }
  }
  p.live[*host_ptr] = nbytes;
  p.in_use += nbytes;
  return PSH_OK;
}

int pinned_free(void *host_ptr) {
  if (!host_ptr) return PSH_OK;
  PinnedPool &p = pool();
  std::lock_guard<std::mutex> g(p.mu);
  auto it = p.live.find(host_ptr);
  if (it == p.live.end()) return fail(PSH_EINVAL, "psh_host_free: pointer was not allocated by psh_host_alloc");
  const size_t nbytes = it->second;
  p.live.erase(it);
  p.in_use -= nbytes;
  if (p.cached + nbytes <= p.limit / 2) {
    p.free_blocks[nbytes].push_back(host_ptr);
    p.cached += nbytes;
  } else {
    (void)hipHostFree(host_ptr);
  }
  return PSH_OK;
}

void pinned_release_cache() {
  PinnedPool &p = pool();
  std::lock_guard<std::mutex> g(p.mu);
  for (auto &kv : p.free_blocks)
    for (void *q : kv.second) (void)hipHostFree(q);
  p.free_blocks.clear();
  p.cached = 0;
  if (g_down_stream) {
    (void)hipStreamDestroy(g_down_stream);
    g_down_stream = nullptr;
  }
}

}  // namespace psh

using psh::ctx;
using psh::fail;

extern "C" {

int psh_host_alloc(void **host_ptr, size_t nbytes) {
  PSH_REQUIRE_INIT();
  if (!host_ptr) return fail(PSH_EINVAL, "psh_host_alloc: NULL out pointer");
  PSH_HIP(hipSetDevice(ctx().device));
  return psh::pinned_alloc(host_ptr, nbytes);
}

int psh_host_free(void *host_ptr) {
  if (!ctx().ready) return PSH_OK;  // interpreter shutdown after psh_shutdown: the pool is gone
  return psh::pinned_free(host_ptr);
}

int psh_semilag_host(const void *precip, const void *velocity, int m, int n, const double *steps, int T,
                     int n_iter, int interp_order, float outval, const double *disp_prev, double *disp_out,
                     void *out, int flags, int *input_status) {
  PSH_REQUIRE_INIT();
  if (input_status) *input_status = 0;
  if (int rc = psh::check_semilag(m, n, T, n_iter, interp_order)) return rc;
  if (!velocity || !steps) return fail(PSH_EINVAL, "semilag: NULL velocity/steps");
  if (precip && !out) return fail(PSH_EINVAL, "semilag: precip given but out is NULL");
  if (!precip && !disp_out) return fail(PSH_EINVAL, "semilag: precip is NULL but no displacement output was given");
  const bool p64 = (flags & PSH_SL_PRECIP_F64) != 0, v64 = (flags & PSH_SL_VELOCITY_F64) != 0;
  const bool o64 = (flags & PSH_SL_OUT_F64) != 0;
  psh::Context &c = ctx();
  psh::Lock lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (!psh::g_down_stream) PSH_HIP(hipStreamCreateWithFlags(&psh::g_down_stream, hipStreamNonBlocking));
  const size_t plane = static_cast<size_t>(m) * n;
  float *d_p = nullptr, *d_v = nullptr, *d_out = nullptr;
  double *d_disp = nullptr;
  void *d_raw = nullptr, *d_out64 = nullptr;  // float64 inputs as uploaded / float64 results before the download
  psh::Ring ring;
  hipEvent_t done = nullptr;
  // device blocks come from the stream-ordered cache: a nowcast loop that calls this entry point
  // once per member and time step does not pay hipMalloc/hipFree each time.  They are handed back
  // only after the download stream has drained (it reads them outside the library stream's order).
  auto cleanup = [&](bool wait) {
    if (wait) {
      lock.unlock();
      (void)hipStreamSynchronize(psh::g_down_stream);
      if (done) (void)hipEventSynchronize(done);
      lock.lock();
    }
    if (done) (void)hipEventDestroy(done);
    ring.destroy();
    for (void *q : {static_cast<void *>(d_p), static_cast<void *>(d_v), static_cast<void *>(d_out),
                    static_cast<void *>(d_disp), d_raw, d_out64})
      if (q) (void)psh_free(q);
  };
#define PSH_TRY_RC(expr)       \
  do {                         \
    const int _rc = (expr);    \
    if (_rc != PSH_OK) {       \
      cleanup(true);           \
      return _rc;              \
    }                          \
  } while (0)
#define PSH_TRY_HIP(expr)                                                                              \
  do {                                                                                                 \
    const hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                            \
      cleanup(true);                                                                                   \
      return fail(_e == hipErrorOutOfMemory ? PSH_ENOMEM : PSH_EHIP, "%s failed: %s", #expr,           \
                  hipGetErrorString(_e));                                                              \
    }                                                                                                  \
  } while (0)
  PSH_TRY_RC(psh_malloc(reinterpret_cast<void **>(&d_v), 2 * plane * sizeof(float)));
  if (precip) {
    PSH_TRY_RC(psh_malloc(reinterpret_cast<void **>(&d_p), plane * sizeof(float)));
    PSH_TRY_RC(psh_malloc(reinterpret_cast<void **>(&d_out), static_cast<size_t>(T) * plane * sizeof(float)));
    if (o64) PSH_TRY_RC(psh_malloc(&d_out64, static_cast<size_t>(T) * plane * sizeof(double)));
  }
  if (disp_prev || disp_out) PSH_TRY_RC(psh_malloc(reinterpret_cast<void **>(&d_disp), 2 * plane * sizeof(double)));
  // float64 arrays (what pysteps' importers produce) cross the bus as they are and are narrowed on
  // the device: a host-side astype(float32) of three 4096^2 planes costs more than the whole call
  if (p64 || v64) PSH_TRY_RC(psh_malloc(&d_raw, 2 * plane * sizeof(double)));
  // float64 inputs are checked as float64, before the narrowing (a finite value beyond the float32
  // range is not a "non-finite input", and outval="min" is the minimum of the values the caller gave)
  psh::FieldStats sv, sp;
  bool have_sv = false, have_sp = false;
  if (v64) {
    PSH_TRY_RC(psh::upload(d_raw, velocity, 2 * plane * sizeof(double), c.stream, ring, lock));
    PSH_TRY_RC(psh::field_stats_full_f64(static_cast<const double *>(d_raw), 2 * plane, &sv));
    have_sv = true;
    PSH_TRY_HIP(psh::launch_convert_f64_f32(static_cast<const double *>(d_raw), d_v, 2 * plane, c.stream));
  } else {
    PSH_TRY_RC(psh::upload(d_v, velocity, 2 * plane * sizeof(float), c.stream, ring, lock));
  }
  if (precip && p64) {
    PSH_TRY_RC(psh::upload(d_raw, precip, plane * sizeof(double), c.stream, ring, lock));
    PSH_TRY_RC(psh::field_stats_full_f64(static_cast<const double *>(d_raw), plane, &sp));
    have_sp = true;
    PSH_TRY_HIP(psh::launch_convert_f64_f32(static_cast<const double *>(d_raw), d_p, plane, c.stream));
  } else if (precip) {
    PSH_TRY_RC(psh::upload(d_p, precip, plane * sizeof(float), c.stream, ring, lock));
  }
  if (disp_prev) PSH_TRY_RC(psh::upload(d_disp, disp_prev, 2 * plane * sizeof(double), c.stream, ring, lock));
  // the input checks of semilagrangian.py:106-137 as device reductions (a NumPy isfinite scan of
  // the three planes costs ~10 ms at 4096^2, these two reductions ~40 us + one stream sync)
  {
    if (!have_sv) PSH_TRY_RC(psh::field_stats_full(d_v, 2 * plane, &sv));
    int status = 0;
    if (sv.nonfinite > 0) status |= PSH_SL_ST_VELOCITY_NONFINITE;
    if (sv.nonfinite >= sv.count) status |= PSH_SL_ST_VELOCITY_ALL_NONFINITE;
    if (precip) {
      if (!have_sp) PSH_TRY_RC(psh::field_stats_full(d_p, plane, &sp));
      if (sp.nonfinite > 0) status |= PSH_SL_ST_PRECIP_NONFINITE;
      if (sp.nonfinite >= sp.count) status |= PSH_SL_ST_PRECIP_ALL_NONFINITE;
      if (flags & PSH_SL_OUTVAL_MIN) outval = static_cast<float>(sp.nanmin());  // :171-172
    }
    if (input_status) *input_status = status;
    const int fatal = PSH_SL_ST_VELOCITY_ALL_NONFINITE | PSH_SL_ST_PRECIP_ALL_NONFINITE |
                      ((flags & PSH_SL_ALLOW_NONFINITE) ? 0 : (PSH_SL_ST_VELOCITY_NONFINITE | PSH_SL_ST_PRECIP_NONFINITE));
    if (status & fatal) {
      cleanup(true);
      return fail(PSH_EINPUT, "semilag: non-finite input values (status 0x%x)", status);
    }
  }
  PSH_TRY_RC(psh_semilag_dev(d_p, d_v, m, n, steps, T, n_iter, interp_order, outval, d_disp,
                             disp_prev == nullptr ? 0 : ((flags & PSH_SL_BASE_IN_DISP) ? PSH_SL_RESUME_BASE : 1), d_out));
  const size_t out_elems = static_cast<size_t>(T) * plane;
  if (precip && o64)
    PSH_TRY_HIP(psh::launch_convert_f32_f64(d_out, static_cast<double *>(d_out64), out_elems, c.stream));
  PSH_TRY_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  PSH_TRY_HIP(hipEventRecord(done, c.stream));
  PSH_TRY_HIP(hipStreamWaitEvent(psh::g_down_stream, done, 0));
  if (precip)
    PSH_TRY_RC(psh::download(out, o64 ? d_out64 : static_cast<void *>(d_out), out_elems * (o64 ? 8 : 4), ring, lock));
  if (disp_out) PSH_TRY_RC(psh::download(disp_out, d_disp, 2 * plane * sizeof(double), ring, lock));
#undef PSH_TRY_RC
#undef PSH_TRY_HIP
  lock.unlock();
  const hipError_t e = hipStreamSynchronize(psh::g_down_stream);
  lock.lock();
  cleanup(false);
  PSH_HIP(e);
  return PSH_OK;
}

}  // extern "C"
