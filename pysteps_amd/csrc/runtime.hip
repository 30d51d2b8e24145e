// Runtime half of the C ABI (include/pysteps_hip.h): device binding, memory,
// events, error reporting.  One process drives one GPU (one rank per device).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "common.h"

namespace psh {

Context &ctx() {
  static Context c;
  return c;
}

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// generation of the scratch allocation: whoever remembers a pointer into it (the step factors of the last
// extrapolation) compares generations, not addresses - a new block can land where the old one was
static unsigned long long g_scratch_generation = 1;

int ensure_scratch(size_t nbytes) {
  Context &c = ctx();
  if (c.scratch_bytes >= nbytes) return PSH_OK;
  ++g_scratch_generation;
  if (c.scratch) {
    PSH_HIP(hipStreamSynchronize(c.stream));
    PSH_HIP(hipFree(c.scratch));
    c.scratch = nullptr;
    c.scratch_bytes = 0;
  }
  size_t want = nbytes < 65536 ? 65536 : nbytes;
  PSH_HIP(hipMalloc(&c.scratch, want));
  c.scratch_bytes = want;
  return PSH_OK;
}

// ---- stream-ordered caching allocator ---------------------------------------
// Every kernel and copy of the library is ordered on ONE stream, so a block handed
// back by psh_free() can be reused by the next psh_malloc() of the same size
// without synchronising: whatever was queued on the old owner runs before anything
// queued on the new one.  This keeps the per-call scratch of the operators
// (1.5 GiB of output planes per 4096^2 x 24 nowcast) off the hipMalloc/hipFree path.
struct BlockCache {
  std::map<size_t, std::vector<void *>> free_blocks;
  std::map<void *, size_t> live;  // size of every block handed out
  size_t cached_bytes = 0;
  size_t limit = 0;
};

static BlockCache &cache() {
  static BlockCache c;
  if (c.limit == 0) {
    const char *env = std::getenv("PYSTEPS_HIP_CACHE_BYTES");
    c.limit = env ? static_cast<size_t>(std::strtoull(env, nullptr, 10)) : (size_t(32) << 30);
    if (c.limit == 0) c.limit = 1;
  }
  return c;
}

static void release_cache() {
  BlockCache &bc = cache();
  for (auto &kv : bc.free_blocks)
    for (void *p : kv.second) (void)hipFree(p);
  bc.free_blocks.clear();
  bc.cached_bytes = 0;
}

static int ensure_pinned(size_t nbytes) {
  Context &c = ctx();
  if (c.pinned_bytes >= nbytes) return PSH_OK;
  if (c.pinned) {
    PSH_HIP(hipStreamSynchronize(c.stream));
    PSH_HIP(hipHostFree(c.pinned));
    c.pinned = nullptr;
    c.pinned_bytes = 0;
  }
  PSH_HIP(hipHostMalloc(&c.pinned, nbytes, hipHostMallocDefault));
  c.pinned_bytes = nbytes;
  return PSH_OK;
}

// A fresh 4 KiB pinned slot + its device twin per call, so that back-to-back asynchronous calls
// never overwrite constants a queued kernel still has to read; the ring synchronises with the
// stream only when it wraps (every 64 calls).
constexpr size_t kConstSlots = 64;

int const_slot(float **host, const float **dev) {
  Context &c = ctx();
  static size_t slot = 0;
  const size_t n_slots = kConstSlots;
  // one more device slot than the ring hands out: the extrapolator's private one (semilag_factor_slot)
  if (int rc = ensure_scratch((n_slots + 1) * kConstSlotFloats * sizeof(float))) return rc;
  if (int rc = ensure_pinned(n_slots * kConstSlotFloats * sizeof(float))) return rc;
  if (slot == n_slots) {  // ring wrapped: make sure the oldest slots are consumed
    PSH_HIP(hipStreamSynchronize(c.stream));
    slot = 0;
  }
  *host = static_cast<float *>(c.pinned) + slot * kConstSlotFloats;
  *dev = static_cast<float *>(c.scratch) + slot * kConstSlotFloats;
  ++slot;
  return PSH_OK;
}

// The device slot behind the ring: only the extrapolator writes it (its per-step scale factors, kept from
// call to call while they do not change).  No ring user can overwrite it however many slots are taken in
// between; it moves only when the scratch block is reallocated (g_scratch_generation).
static int semilag_factor_slot(const float **dev) {
  if (int rc = ensure_scratch((kConstSlots + 1) * kConstSlotFloats * sizeof(float))) return rc;
  *dev = static_cast<float *>(ctx().scratch) + kConstSlots * kConstSlotFloats;
  return PSH_OK;
}

int check_semilag(int m, int n, int T, int n_iter, int order_and_mode) {
  const int order = order_and_mode & 0xff, bmode = (order_and_mode >> 8) & 0xff;
  if (order_and_mode < 0 || (order_and_mode >> 16) != 0 || bmode > PSH_MODE_GRID_WRAP)
    return fail(PSH_EINVAL, "semilag: invalid interp_order / boundary mode word 0x%x", order_and_mode);
  if (m <= 0 || n <= 0) return fail(PSH_EINVAL, "semilag: invalid shape (%d,%d)", m, n);
  if (static_cast<uint64_t>(m) * static_cast<uint64_t>(n) >= (1ull << 30))
    return fail(PSH_EUNSUPPORTED, "semilag: m*n must be < 2^30 pixels (32-bit byte offsets)");
  if (T <= 0) return fail(PSH_EINVAL, "semilag: T must be positive (got %d)", T);
  if (n_iter < 0) return fail(PSH_EINVAL, "semilag: n_iter must be >= 0 (got %d)", n_iter);
  if (order < 0 || order > 5)
    return fail(PSH_EUNSUPPORTED, "semilag: interp_order %d not implemented (0 .. 5)", order);
  return PSH_OK;
}

}  // namespace psh

using psh::ctx;
using psh::fail;

extern "C" {

const char *psh_last_error(void) { return psh::g_err; }

const char *psh_version(void) { return "pysteps_hip 0.1 (gfx950)"; }

int psh_init(int device_id) {
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  if (c.ready) {
    if (device_id >= 0 && device_id != c.device)
      return fail(PSH_EINVAL, "already bound to device %d (asked for %d)", c.device, device_id);
    return PSH_OK;
  }
  int count = 0;
  PSH_HIP(hipGetDeviceCount(&count));
  if (count <= 0) return fail(PSH_EHIP, "no HIP device visible");
  if (device_id < 0) device_id = 0;
  if (device_id >= count)
    return fail(PSH_EINVAL, "device %d out of range (%d visible)", device_id, count);
  PSH_HIP(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  PSH_HIP(hipGetDeviceProperties(&prop, device_id));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PSH_EUNSUPPORTED, "device %d is %s; this library is built for gfx950 only",
                device_id, prop.gcnArchName);
  c.device = device_id;
  c.cu_count = prop.multiProcessorCount;
  PSH_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  c.ready = true;
  return PSH_OK;
}

extern "C++" {
namespace psh {
namespace {
struct PinnedSlot {
  void **slot;
};
std::vector<PinnedSlot> &pinned_slots() {
  static std::vector<PinnedSlot> v;
  return v;
}
}  // namespace

int persistent_pinned(void **slot, size_t nbytes) {
  if (*slot) return PSH_OK;
  PSH_HIP(hipHostMalloc(slot, nbytes, hipHostMallocDefault));
  pinned_slots().push_back(PinnedSlot{slot});
  return PSH_OK;
}

static void release_persistent_pinned() {
  for (PinnedSlot &p : pinned_slots()) {
    if (*p.slot) (void)hipHostFree(*p.slot);
    *p.slot = nullptr;
  }
  pinned_slots().clear();
}

int side_begin(hipStream_t *side) {
  Context &c = ctx();
  if (!c.side) {
    PSH_HIP(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
    PSH_HIP(hipEventCreateWithFlags(&c.fork_ev, hipEventDisableTiming));
    PSH_HIP(hipEventCreateWithFlags(&c.join_ev, hipEventDisableTiming));
  }
  PSH_HIP(hipEventRecord(c.fork_ev, c.stream));
  PSH_HIP(hipStreamWaitEvent(c.side, c.fork_ev, 0));
  *side = c.side;
  return PSH_OK;
}
int side_end() {
  Context &c = ctx();
  if (!c.side) return PSH_OK;
  PSH_HIP(hipEventRecord(c.join_ev, c.side));
  PSH_HIP(hipStreamWaitEvent(c.stream, c.join_ev, 0));
  return PSH_OK;
}
}  // namespace psh
}  // extern "C++"

int psh_shutdown(void) {
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  if (!c.ready) return PSH_OK;
  (void)hipStreamSynchronize(c.stream);
  psh::release_cache();
  psh::pinned_release_cache();
  psh::release_persistent_pinned();
  psh::fft_release();
  if (c.scratch) (void)hipFree(c.scratch);
  if (c.mask_any) (void)hipFree(c.mask_any);
  c.mask_any = nullptr;
  ++psh::g_scratch_generation;
  if (c.pinned) (void)hipHostFree(c.pinned);
  (void)hipStreamDestroy(c.stream);
  if (c.side) {
    (void)hipStreamSynchronize(c.side);
    (void)hipStreamDestroy(c.side);
    (void)hipEventDestroy(c.fork_ev);
    (void)hipEventDestroy(c.join_ev);
    c.side = nullptr;
    c.fork_ev = c.join_ev = nullptr;
  }
  c.scratch = c.pinned = nullptr;
  c.scratch_bytes = c.pinned_bytes = 0;
  c.stream = nullptr;
  c.ready = false;
  return PSH_OK;
}

int psh_set_option(const char *key, int value) {
  if (!key) return fail(PSH_EINVAL, "psh_set_option: NULL key");
  if (std::strcmp(key, "semilag_variant") == 0) {
    if (value != 0 && value != 1 && value != 5 && value != 7 && value != 12)
      return fail(PSH_EINVAL,
                  "semilag_variant must be 0 (window kernel where it applies, gather kernels elsewhere), 12 (window kernel "
                  "for every eligible call), 7 (gather kernels: packed {u,v} plane + row-pair field plane), 5 (packed "
                  "velocity only) or 1 (one plane per component, DPP column sharing)");
    psh::set_semilag_variant(value);
    return PSH_OK;
  }
  if (std::strcmp(key, "members_variant") == 0) {
    if (value != 1 && value != 2)
      return fail(PSH_EINVAL, "members_variant must be 2 (two members per thread, default) or 1 (one)");
    psh::set_members_variant(value);
    return PSH_OK;
  }
  if (std::strcmp(key, "idw_variant") == 0) {
    if (value < 0 || value > 1) return fail(PSH_EINVAL, "idw_variant must be 0 (two-level) or 1 (pre-pass per 16x16 tile)");
    psh::set_idw_variant(value);
    return PSH_OK;
  }
  if (std::strcmp(key, "lk_fused_nms") == 0) {
    if (value != 0 && value != 1) return fail(PSH_EINVAL, "lk_fused_nms must be 0 (response + selection passes, default) or 1 (one fused pass)");
    psh::set_lk_fused_nms(value);
    return PSH_OK;
  }
  if (std::strcmp(key, "trim_cache") == 0) {  // give the cached device blocks back to the driver
    psh::Context &c = ctx();
    std::lock_guard<std::recursive_mutex> lock(c.mu);
    if (c.ready) {
      PSH_HIP(hipSetDevice(c.device));
      PSH_HIP(hipStreamSynchronize(c.stream));
      psh::release_cache();
      psh::pinned_release_cache();
    }
    return PSH_OK;
  }
  return fail(PSH_EINVAL, "psh_set_option: unknown option '%s'", key);
}

int psh_field_stats_dev(const float *in_dev, size_t n, double *min_out, double *max_out,
                        double *nonfinite_out);

int psh_device_info(int *device_id, int *cu_count, size_t *hbm_total, size_t *hbm_free,
                    char *name, int name_len) {
  PSH_REQUIRE_INIT();
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  if (device_id) *device_id = c.device;
  if (cu_count) *cu_count = c.cu_count;
  if (hbm_total || hbm_free) {
    size_t f = 0, t = 0;
    PSH_HIP(hipMemGetInfo(&f, &t));
    if (hbm_total) *hbm_total = t;
    if (hbm_free) *hbm_free = f;
  }
  if (name && name_len > 0) {
    hipDeviceProp_t prop;
    PSH_HIP(hipGetDeviceProperties(&prop, c.device));
    std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  return PSH_OK;
}

int psh_malloc(void **dev_ptr, size_t nbytes) {
  PSH_REQUIRE_INIT();
  if (!dev_ptr) return fail(PSH_EINVAL, "psh_malloc: NULL out pointer");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  *dev_ptr = nullptr;
  if (nbytes == 0) return PSH_OK;
  nbytes = (nbytes + 255) & ~static_cast<size_t>(255);
  psh::BlockCache &bc = psh::cache();
  auto it = bc.free_blocks.find(nbytes);
  if (it != bc.free_blocks.end() && !it->second.empty()) {
    *dev_ptr = it->second.back();
    it->second.pop_back();
    bc.cached_bytes -= nbytes;
    bc.live[*dev_ptr] = nbytes;
    return PSH_OK;
  }
  hipError_t e = hipMalloc(dev_ptr, nbytes);
  if (e == hipErrorOutOfMemory) {  // give the cached blocks back and retry once
    (void)hipGetLastError();
    (void)hipStreamSynchronize(c.stream);
    psh::release_cache();
    e = hipMalloc(dev_ptr, nbytes);
  }
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    return fail(PSH_ENOMEM, "hipMalloc(%zu) out of device memory", nbytes);
  }
  PSH_HIP(e);
  bc.live[*dev_ptr] = nbytes;
  return PSH_OK;
}

int psh_free(void *dev_ptr) {
  PSH_REQUIRE_INIT();
  if (!dev_ptr) return PSH_OK;
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::BlockCache &bc = psh::cache();
  auto it = bc.live.find(dev_ptr);
  if (it == bc.live.end()) return fail(PSH_EINVAL, "psh_free: pointer was not allocated by psh_malloc");
  const size_t nbytes = it->second;
  bc.live.erase(it);
  if (bc.cached_bytes + nbytes <= bc.limit) {  // stream-ordered reuse, no synchronisation
    bc.free_blocks[nbytes].push_back(dev_ptr);
    bc.cached_bytes += nbytes;
    return PSH_OK;
  }
  PSH_HIP(hipStreamSynchronize(c.stream));
  PSH_HIP(hipFree(dev_ptr));
  return PSH_OK;
}

int psh_memcpy_h2d(void *dst_dev, const void *src_host, size_t nbytes) {
  PSH_REQUIRE_INIT();
  if (nbytes == 0) return PSH_OK;
  if (!dst_dev || !src_host) return fail(PSH_EINVAL, "psh_memcpy_h2d: NULL pointer");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  // pageable source: the runtime stages it, after which the host buffer may be reused
  PSH_HIP(hipMemcpyAsync(dst_dev, src_host, nbytes, hipMemcpyHostToDevice, c.stream));
  return PSH_OK;
}

int psh_memcpy_d2h(void *dst_host, const void *src_dev, size_t nbytes) {
  PSH_REQUIRE_INIT();
  if (nbytes == 0) return PSH_OK;
  if (!dst_host || !src_dev) return fail(PSH_EINVAL, "psh_memcpy_d2h: NULL pointer");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipMemcpyAsync(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost, c.stream));
  PSH_HIP(hipStreamSynchronize(c.stream));
  return PSH_OK;
}

int psh_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t nbytes) {
  PSH_REQUIRE_INIT();
  if (nbytes == 0) return PSH_OK;
  if (!dst_host || !src_dev) return fail(PSH_EINVAL, "psh_memcpy_d2h_async: NULL pointer");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipMemcpyAsync(dst_host, src_dev, nbytes, hipMemcpyDeviceToHost, c.stream));
  return PSH_OK;
}

int psh_memcpy_d2d(void *dst_dev, const void *src_dev, size_t nbytes) {
  PSH_REQUIRE_INIT();
  if (nbytes == 0) return PSH_OK;
  if (!dst_dev || !src_dev) return fail(PSH_EINVAL, "psh_memcpy_d2d: NULL pointer");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipMemcpyAsync(dst_dev, src_dev, nbytes, hipMemcpyDeviceToDevice, c.stream));
  return PSH_OK;
}

int psh_memset(void *dst_dev, int byte_value, size_t nbytes) {
  PSH_REQUIRE_INIT();
  if (nbytes == 0) return PSH_OK;
  if (!dst_dev) return fail(PSH_EINVAL, "psh_memset: NULL pointer");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipMemsetAsync(dst_dev, byte_value, nbytes, c.stream));
  return PSH_OK;
}

int psh_sync(void) {
  PSH_REQUIRE_INIT();
  psh::Context &c = ctx();
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipStreamSynchronize(c.stream));
  return PSH_OK;
}

int psh_event_create(void **event) {
  PSH_REQUIRE_INIT();
  if (!event) return fail(PSH_EINVAL, "psh_event_create: NULL out pointer");
  PSH_HIP(hipSetDevice(ctx().device));
  hipEvent_t e;
  PSH_HIP(hipEventCreate(&e));
  *event = e;
  return PSH_OK;
}

int psh_event_destroy(void *event) {
  PSH_REQUIRE_INIT();
  if (event) PSH_HIP(hipEventDestroy(static_cast<hipEvent_t>(event)));
  return PSH_OK;
}

int psh_event_record(void *event) {
  PSH_REQUIRE_INIT();
  if (!event) return fail(PSH_EINVAL, "psh_event_record: NULL event");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  PSH_HIP(hipEventRecord(static_cast<hipEvent_t>(event), c.stream));
  return PSH_OK;
}

int psh_event_elapsed_ms(void *start, void *stop, float *ms) {
  PSH_REQUIRE_INIT();
  if (!start || !stop || !ms) return fail(PSH_EINVAL, "psh_event_elapsed_ms: NULL argument");
  PSH_HIP(hipSetDevice(ctx().device));
  PSH_HIP(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  PSH_HIP(hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
  return PSH_OK;
}

// ---------------------------------------------------------------------------
// semi-Lagrangian extrapolation
// ---------------------------------------------------------------------------

static int semilag_rows(const float *precip_dev, const float *velocity_dev, const float *velocity_uv_dev, int m, int n,
                        const double *steps_host, int T, int n_iter, int interp_order, float outval, double *disp_dev,
                        int resume, int row_begin, int row_count, float *out_dev);

int psh_semilag_window_shape(int m, int n) { return psh::semilag_window_shape(m, n) ? 1 : 0; }

int psh_semilag_kernel(int m, int n, int T, int n_iter, int interp_order, int has_field) {
  return psh::semilag_kernel_choice(m, n, T, n_iter, interp_order, has_field != 0);
}

int psh_semilag_rows_dev(const float *precip_dev, const float *velocity_dev, int m, int n,
                         const double *steps_host, int T, int n_iter, int interp_order,
                         float outval, double *disp_dev, int resume, int row_begin, int row_count,
                         float *out_dev) {
  return semilag_rows(precip_dev, velocity_dev, nullptr, m, n, steps_host, T, n_iter, interp_order, outval, disp_dev, resume,
                      row_begin, row_count, out_dev);
}

// velocity_uv_dev (may be NULL): the {u, v}-interleaved (m, n, 2) copy of velocity_dev that the kernel gathers from,
// given by the caller (psh_dense_lk_uv_dev writes one) instead of being made by a pass over the planes on every call
int psh_semilag_uv_dev(const float *precip_dev, const float *velocity_dev, const float *velocity_uv_dev, int m, int n,
                       const double *steps_host, int T, int n_iter, int interp_order, float outval, double *disp_dev,
                       int resume, float *out_dev) {
  return semilag_rows(precip_dev, velocity_dev, velocity_uv_dev, m, n, steps_host, T, n_iter, interp_order, outval, disp_dev,
                      resume, 0, m, out_dev);
}

int psh_semilag_dev(const float *precip_dev, const float *velocity_dev, int m, int n,
                    const double *steps_host, int T, int n_iter, int interp_order,
                    float outval, double *disp_dev, int resume, float *out_dev) {
  return psh_semilag_rows_dev(precip_dev, velocity_dev, m, n, steps_host, T, n_iter, interp_order,
                              outval, disp_dev, resume, 0, m, out_dev);
}

static int semilag_rows(const float *precip_dev, const float *velocity_dev, const float *velocity_uv_dev, int m, int n,
                        const double *steps_host, int T, int n_iter, int interp_order, float outval, double *disp_dev,
                        int resume, int row_begin, int row_count, float *out_dev) {
  PSH_REQUIRE_INIT();
  if (row_begin < 0 || row_count <= 0 || row_begin + row_count > m)
    return fail(PSH_EINVAL, "semilag: row band [%d, %d) outside the %d-row image", row_begin,
                row_begin + row_count, m);
  if (int rc = psh::check_semilag(m, n, T, n_iter, interp_order)) return rc;
  if (!velocity_dev || !steps_host) return fail(PSH_EINVAL, "semilag: NULL velocity/steps");
  if (precip_dev && !out_dev) return fail(PSH_EINVAL, "semilag: precip given but out is NULL");
  if (!precip_dev && !disp_dev)
    return fail(PSH_EINVAL, "semilag: precip is NULL but no displacement buffer was given");
  if (resume && !disp_dev) return fail(PSH_EINVAL, "semilag: resume without displacement");
  psh::Context &c = ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  // per-step scale factors through the pinned slot ring (no host synchronisation)
  if (static_cast<size_t>(T) > psh::kConstSlotFloats)
    return fail(PSH_EUNSUPPORTED, "semilag: at most %zu lead steps per call", psh::kConstSlotFloats);
  // ... unless they are the factors of the previous call (a nowcast advects by the same increments call after
  // call): they live in a device slot of their own that no other entry point writes (the ring's slots are
  // recycled after 64 calls of ANY ring user: masks, cascades, member loops), so the copy with its dispatch gap
  // (~10 us in front of the kernel) is skipped.  A changed vector overwrites the slot in stream order: every
  // kernel that read the old factors was queued before the copy.
  static std::vector<float> last_scales;
  static const float *last_dev = nullptr;
  static unsigned long long last_generation = 0;
  const double sub = n_iter > 1 ? static_cast<double>(n_iter) : 1.0;
  std::vector<float> scales(static_cast<size_t>(T));
  for (int t = 0; t < T; ++t) scales[static_cast<size_t>(t)] = static_cast<float>(steps_host[t] / sub);
  const float *d = nullptr;
  if (last_dev != nullptr && last_generation == psh::g_scratch_generation && scales.size() == last_scales.size() &&
      std::memcmp(scales.data(), last_scales.data(), scales.size() * sizeof(float)) == 0) {
    d = last_dev;
  } else {
    float *h = nullptr;
    const float *ring_dev = nullptr;  // only the pinned half of the ring slot is used (staging)
    if (int rc = psh::const_slot(&h, &ring_dev)) return rc;
    if (int rc = psh::semilag_factor_slot(&d)) return rc;
    std::memcpy(h, scales.data(), scales.size() * sizeof(float));
    PSH_HIP(hipMemcpyAsync(const_cast<float *>(d), h, T * sizeof(float), hipMemcpyHostToDevice, c.stream));
    last_scales = scales;
    last_dev = d;
    last_generation = psh::g_scratch_generation;
  }

  psh::SemilagArgs a;
  a.precip = precip_dev;
  a.vel = velocity_dev;
  a.out = out_dev;
  a.disp = disp_dev;
  a.scale = d;
  a.first_scale = static_cast<float>(steps_host[0]);
  a.m = m;
  a.n = n;
  a.T = T;
  a.n_iter = n_iter;
  a.order = interp_order & 0xff;
  a.bmode = (interp_order >> 8) & 0xff;
  interp_order &= 0xff;
  if (interp_order >= 2 && interp_order <= 5) {  // B-spline resampling: one kernel instantiation, the order as data
    a.spline_order = interp_order;
    a.order = interp_order = 3;
  }
  a.resume = resume;
  a.row0 = row_begin;
  a.rows = row_count;
  a.outval = outval;
  a.coef = nullptr;
  a.minval = 0.f;
  void *spline_blk = nullptr;
  if (interp_order == 3 && precip_dev) {
    // cubic B-spline coefficients (spline.hip) + the minimum the mask logic restores (:146-147).  Per boundary
    // mode: the filter's boundary kind, and for the two modes SciPy pads before filtering the padded plane
    const int mode = a.bmode;
    const int kind = (mode == PSH_MODE_NEAREST || mode == PSH_MODE_REFLECT) ? 1 : (mode == PSH_MODE_GRID_WRAP ? 2 : 0);
    const int npad = (mode == PSH_MODE_NEAREST || mode == PSH_MODE_GRID_CONSTANT) ? 12 : 0;
    // a non-finite cval padded around the field reaches every coefficient through the filter's recursions
    const bool all_nan = mode == PSH_MODE_GRID_CONSTANT && !std::isfinite(outval);
    const size_t plane_bytes = static_cast<size_t>(m + 2 * npad) * (n + 2 * npad) * sizeof(float);
    hipError_t e = hipSuccess;
    if (!all_nan) {
      if (int rc = psh_malloc(&spline_blk, 2 * plane_bytes)) return rc;
      float *coef = static_cast<float *>(spline_blk);
      float *tmp = coef + plane_bytes / sizeof(float);
      e = psh::spline_prefilter(precip_dev, coef, tmp, m, n, c.stream, kind, npad, mode == PSH_MODE_NEAREST, outval,
                                a.spline_order);
      a.coef = coef;
      a.coef_pad = npad;
    }
    double mn = 0.0;
    int rc = e == hipSuccess ? psh_field_stats_dev(precip_dev, static_cast<size_t>(m) * n, &mn, nullptr, nullptr)
                             : fail(PSH_EHIP, "spline prefilter failed: %s", hipGetErrorString(e));
    if (rc) {
      if (spline_blk) (void)psh_free(spline_blk);
      return rc;
    }
    a.minval = static_cast<float>(mn);
  }
  void *packed_blk = nullptr;
  if (psh::semilag_wants_packed(a)) {
    // {u,v} interleaved copy of the velocity for the dwordx4 gathers (semilag.hip); cached block
    // and, for bilinear resampling over several lead times, the row-pair copy of the field behind it
    const size_t plane = static_cast<size_t>(m) * n;
    const bool pairs = psh::semilag_wants_field_pairs(a);
    const bool own_uv = velocity_uv_dev == nullptr || reinterpret_cast<uintptr_t>(velocity_uv_dev) % 16 != 0;
    int rc = PSH_OK;
    if (own_uv || pairs) rc = psh_malloc(&packed_blk, ((own_uv ? 2 : 0) + (pairs ? 2 : 0)) * plane * sizeof(float));
    if (rc == PSH_OK) {
      hipError_t pe = hipSuccess;
      if (own_uv) pe = psh::launch_pack_velocity(velocity_dev, static_cast<float *>(packed_blk), plane, c.stream);
      if (pe == hipSuccess && pairs) {
        float *pp = static_cast<float *>(packed_blk) + (own_uv ? 2 : 0) * plane;
        pe = psh::launch_pack_field_rows(precip_dev, pp, m, n, c.stream);
        a.field_pairs = pp;
      }
      if (pe != hipSuccess) rc = fail(PSH_EHIP, "pack_velocity failed: %s", hipGetErrorString(pe));
    }
    if (rc) {
      if (packed_blk) (void)psh_free(packed_blk);
      if (spline_blk) (void)psh_free(spline_blk);
      return rc;
    }
    a.vel_packed = own_uv ? static_cast<const float *>(packed_blk) : velocity_uv_dev;
  }
  const hipError_t le = psh::launch_semilag(a, c.stream);
  if (spline_blk) (void)psh_free(spline_blk);  // stream-ordered: the launch above is queued first
  if (packed_blk) (void)psh_free(packed_blk);
  PSH_HIP(le);
  return PSH_OK;
}

}  // extern "C"
