// Shared host-side plumbing of libpysteps_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <mutex>

#include "../../include/pysteps_hip.h"

// Device-side assertions of the debug build (python -m pysteps_amd.build --debug: -O1 -g -DPSH_DEBUG): compiled out
// of the product.  A violated one prints file:line and traps the kernel; the host sees the launch fail.
#ifdef PSH_DEBUG
#include <cassert>
#define PSH_DASSERT(cond) assert(cond)
#else
#define PSH_DASSERT(cond) ((void)0)
#endif

namespace psh {

constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

struct Context {
  bool ready = false;
  int device = -1;
  int cu_count = 0;
  hipStream_t stream = nullptr;
  // second stream for work that is independent of what the main stream does next (side_begin /
  // side_end fork and join it with events; everything else is ordered on `stream`)
  hipStream_t side = nullptr;
  hipEvent_t fork_ev = nullptr, join_ev = nullptr;
  // small persistent device scratch (per-step scale factors etc.)
  void *scratch = nullptr;
  size_t scratch_bytes = 0;
  // "a wet pixel was seen" word of psh_steps_incremental_mask_dev and the generation number its launches stamp it with
  int *mask_any = nullptr;
  int mask_generation = 0;
  // pinned staging buffer for pageable host copies
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  std::recursive_mutex mu;
};

Context &ctx();
int fail(int code, const char *fmt, ...);
int ensure_scratch(size_t nbytes);
// Small pinned host blocks that live as long as the library (staging slots of single translation
// units): allocated on first use, released by psh_shutdown.  *slot stays NULL on failure.
int persistent_pinned(void **slot, size_t nbytes);
// Fork: *side will run after everything queued on the main stream so far.  Join: the main stream
// continues after everything queued on the side stream.  Lock held by the caller.  Blocks handed
// out by psh_malloc are ordered on the MAIN stream: a block used on the side stream has to be
// allocated before the fork and freed after the join.
int side_begin(hipStream_t *side);
int side_end();
// pinned staging slot (kConstSlotFloats floats) and its device twin for small per-call constants;
// the caller fills *host and queues the copy to *dev on the library stream (lock held)
constexpr size_t kConstSlotFloats = 1024;
int const_slot(float **host, const float **dev);

#define PSH_HIP(expr)                                                            \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess)                                                        \
      return ::psh::fail(PSH_EHIP, "%s failed: %s (%s:%d)", #expr,               \
                         hipGetErrorString(_e), __FILE__, __LINE__);             \
  } while (0)

#define PSH_REQUIRE_INIT()                                                       \
  do {                                                                           \
    if (!::psh::ctx().ready)                                                     \
      return ::psh::fail(PSH_ENOTINIT, "psh_init() has not been called");        \
  } while (0)

int check_semilag(int m, int n, int T, int n_iter, int order_and_mode);

// ---- element-wise helpers (elementwise.hip) ---------------------------------
struct FieldStats {
  double min_finite = INFINITY, max_finite = -INFINITY;  // over finite values
  double nonfinite = 0, neg_inf = 0, pos_inf = 0, count = 0;
  // np.nanmin: NaNs ignored, infinities are values
  double nanmin() const {
    if (neg_inf > 0) return -INFINITY;
    if (nonfinite < count) return min_finite;
    return pos_inf > 0 ? INFINITY : NAN;
  }
};
int field_stats_full(const float *in_dev, size_t n, FieldStats *st);  // waits for the library stream; lock held
int field_stats_full_f64(const double *in_dev, size_t n, FieldStats *st);
hipError_t launch_convert_f64_f32(const double *in, float *out, size_t n, hipStream_t stream);
hipError_t launch_convert_f32_f64(const float *in, double *out, size_t n, hipStream_t stream);

// ---- host-buffer path (hostpath.hip): pinned block pool, staged transfers ---
// pinned host blocks, cached like the device blocks; PSH_ENOMEM beyond PYSTEPS_HIP_PINNED_BYTES
int pinned_alloc(void **host_ptr, size_t nbytes);
int pinned_free(void *host_ptr);
void pinned_release_cache();

// ---- kernel launchers (one per .hip translation unit) ----------------------
struct SemilagArgs {
  const float *precip;  // (m,n) or nullptr
  const float *vel;     // (2,m,n)
  float *out;           // (T,m,n) or nullptr
  double *disp;         // (2,m,n) in/out or nullptr
  const float *scale;   // device, T floats: step / vel_timestep / max(n_iter, 1)
  float first_scale;    // step[0] / vel_timestep (the very first increment is not divided)
  int m, n, T, n_iter, order, resume;
  int row0, rows;       // output row band [row0, row0+rows); out is (T,rows,n)
  float outval;
  const float *coef;    // interp_order 3: cubic B-spline coefficients of precip (m + 2 coef_pad, n + 2 coef_pad); nullptr
                        // with bmode != 0: every coefficient is NaN (a non-finite cval was padded in)
  int coef_pad = 0;     // samples the coefficient plane is padded by ("nearest", "grid-constant": 12)
  float minval;         // interp_order 3: minimum over the finite values of precip
  int bmode = 0;        // boundary mode of the field resampling (PSH_MODE_*), interp_order 0/1
  int spline_order = 3; // order == 3 stands for "B-spline resampling": 2, 3, 4 or 5 (interp_order of the call)
  const float *vel_packed = nullptr;  // (m,n,2) {u,v} interleaved copy of vel (launch_pack_velocity) or nullptr
  const float *field_pairs = nullptr;  // (m,n,2) {p(y,x), p(y+1,x)} row-pair copy of precip (launch_pack_field_rows)
};
// true if the default kernel samples the velocity from the packed {u,v} plane for this call
bool semilag_wants_packed(const SemilagArgs &a);
hipError_t launch_pack_velocity(const float *vel, float *uv, size_t plane, hipStream_t stream);
bool semilag_wants_field_pairs(const SemilagArgs &a);
hipError_t launch_pack_field_rows(const float *precip, float *pairs, int m, int n, hipStream_t stream);
hipError_t launch_semilag(const SemilagArgs &a, hipStream_t stream);
bool semilag_window_shape(int m, int n);
int semilag_kernel_choice(int m, int n, int T, int n_iter, int order, bool has_field);
void set_semilag_variant(int v);
void set_members_variant(int v);
void set_lk_fused_nms(int v);
hipError_t spline_prefilter(const float *precip, float *coef, float *tmp, int m, int n, hipStream_t stream, int kind = 0,
                            int npad = 0, int pad_edge = 0, float cval = 0.f, int order = 3);

// sample count and interpolator preamble kept in device memory (written by vectors_finish,
// lk_sparse.hip): the IDW kernels read L / reach from here instead of their launch arguments, so
// that the host never has to know how many vectors survived
struct IdwDyn {
  int L;        // samples in the list
  int mode;     // 0: interpolate; 1: constant field (cu, cv) - no / one sample, all values equal
  float cu, cv;
  float reach;  // farthest a sample can be from a grid node (histogram range of the coarse pass)
  int pad[3];
};

// ---- FFTs (fft.hip) -----------------------------------------------------------
bool fft_shape_supported(int m, int n);
void fft_release();  // frees the twiddle tables (psh_shutdown)
int fft_irfft2_weighted(const void *spec_dev, const double *weights_dev, int m, int n, double *out_dev,
                        void *scratch_dev, unsigned long long *min_key_dev = nullptr);

struct IdwArgs {
  const float *xy;  // (L,2) device: x, y of the sparse vectors
  const float *uv;  // (L,2) device: values
  float *out;       // (2,m,n) device
  float *out_uv = nullptr;  // (m,n,2) device, optional: the same field as {u,v} pairs (two-level kernels only)
  int L, k, m, n;
  float x0, dx, y0, dy;  // target grid: x = x0 + dx*i (i<n), y = y0 + dy*j (j<m)
  float inv_res, power, offset, dmax;
  void *scratch = nullptr;  // idw_scratch_bytes(m, n) of device memory: supertile candidate lists
  // device-resident sample count: L is then the CAPACITY of xy / uv, k the requested neighbour
  // count (not yet clamped to the sample count), dmax is ignored
  const IdwDyn *dyn = nullptr;
};
hipError_t launch_idw(const IdwArgs &a, hipStream_t stream);
int idw_resident(const float *xy_dev, const float *values_dev, int capacity, const IdwDyn *dyn_dev, int m, int n,
                 int k, double power, double dist_offset, float *out_dev, float *out_uv_dev = nullptr);
size_t idw_scratch_bytes(int m, int n);
void set_idw_variant(int v);

hipError_t launch_outliers_pooled(const double *xy_dev, const double *uv_dev, const int *count_dev,
                                  int capacity, int k, double thr, unsigned char *flags_dev,
                                  hipStream_t stream);
// tracker with device-side pooling of the successful vectors (lk.hip); asynchronous.  The points
// come either from the host (staged in a pinned slot that stays valid until the stream has
// consumed it) or from device memory together with their count (points_dev, npts_dev; npts is
// then the capacity).
int lk_track_pool(void *pyramid_handle, const float *points_host, const float *points_dev, const int *npts_dev,
                  int npts, int max_count, double epsilon, double min_eig_threshold, double *pool_xy_dev,
                  double *pool_uv_dev, int *pool_count_dev, int pool_capacity);

// corner candidates of one frame -> accepted corners in goodFeaturesToTrack's order, everything
// on the library stream and in device memory (lk.hip): points_dev holds max_corners (x, y) pairs
// before_walk (may be NULL) is called once everything up to the ordered walk is queued: the walk
// is a single workgroup, so independent work forked there (side_begin) runs beside it.
// walk_stats_host (may be NULL, else int[13]): waits for the stream and returns {chunks, candidates, rounds}
// of the walk and the microseconds it spent {loading, sorting, on coordinates, in block tests, in batches, in all}.
int lk_corners_resident(const unsigned char *feature_u8_dev, const float *clean_dev, float *stats_dev, int m, int n,
                        int block_size, int buffer_mask, double quality_level, double min_distance, int max_corners,
                        float *points_dev, int *npoints_dev, int (*before_walk)(void *) = nullptr,
                        void *before_walk_arg = nullptr, int *walk_stats_host = nullptr,
                        unsigned *slots_cleared = nullptr);

// the three frame passes of dense Lucas-Kanade (lk.hip) on a stream of the caller's choice, with the
// caller's workspace of lk_prepare_ws_bytes(m, n, f64) bytes (lock held)
size_t lk_prepare_ws_bytes(int m, int n, bool f64);
int lk_prepare_on(hipStream_t stream, void *ws, const void *frame_dev, bool f64, int m, int n, int size_opening,
                  int buffer_mask, float *clean_dev, unsigned char *track_u8_dev, unsigned char *feature_u8_dev,
                  float *stats_dev, unsigned *slots_cleared = nullptr, unsigned long long *keepbits = nullptr);
// clean_dev == nullptr (float32 frames only): the cleaned frame is not stored; keepbits (lk_keepbits_bytes(m, n)
// of device memory) receives one bit per pixel - "keeps its value" (finite and not removed by the opening) - and
// the renderings are made from the frame and these bits.  The corner passes then take the FRAME as clean_dev:
// they only read its NaN pattern, which the cleaning does not change
size_t lk_keepbits_bytes(int m, int n);
// Statistic slots of one frame (lk.hip): lk_slot_bytes() of zero-filled device memory that the frame passes and
// the corner passes of the SAME frame fold their minima / maxima into instead of going through a
// single-workgroup finishing kernel; a caller that passes none gets them cleared by a fill launch
size_t lk_slot_bytes();
int lk_pyramids_beside(const unsigned char *prev_u8_dev, const unsigned char *next_u8_dev, int m, int n, int win_w,
                       int win_h, int max_level, void **handle_out);
int lk_pyramids_on_side(hipStream_t side, void *block, size_t block_bytes, const unsigned char *prev_u8_dev,
                        const unsigned char *next_u8_dev, int m, int n, int win_w, int win_h, int max_level,
                        void **handle_out);
size_t lk_pyramids_bytes(int m, int n, int win_w, int win_h, int max_level);

// ordered min-distance acceptance and the post-outlier-test stage on the device (lk_sparse.hip)
int corner_order_max_corners();
bool corner_order_supported(int m, int n, double min_distance, int max_corners);
size_t corner_order_ws_bytes();  // device scratch of one ordering (histogram, header, head keys)
hipError_t launch_corner_order(const unsigned long long *raw_dev, const int *raw_count_dev, int cap,
                               const float *eig_max_dev, float quality, int n, double min_distance,
                               int max_corners, void *ws_dev, float *points_dev, int *npoints_dev,
                               hipStream_t stream, int (*before_walk)(void *), void *before_walk_arg,
                               bool ws_is_cleared, const unsigned *eig_slots_dev = nullptr, int count_bias = 0,
                               float *eig_max_out_dev = nullptr);
size_t corner_order_clear_bytes();  // leading bytes of the workspace that have to be zero (histogram, header)
size_t corner_order_walk_stats_offset();  // byte offset of the int[3] walk statistics inside the workspace
hipError_t launch_vectors_finish(const double *pool_xy_dev, const double *pool_uv_dev,
                                 const unsigned char *flags_dev, const int *pool_count_dev, int capacity,
                                 double decl_scale, int m, int n, float *xy_out_dev, float *uv_out_dev,
                                 IdwDyn *dyn_dev, hipStream_t stream);

}  // namespace psh
