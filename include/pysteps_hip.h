/* pysteps_hip.h - C ABI of libpysteps_hip.so (MI355X / gfx950 advection hot path).
 *
 * pysteps has no FFI of its own for this path: the operators are plain Python
 * callables looked up by name in two module-level tables,
 *     pysteps/motion/interface.py:36-46         (_methods -> dense_lucaskanade)
 *     pysteps/extrapolation/interface.py:107-111 (_extrapolation_methods -> extrapolate)
 * The drop-in boundary is therefore "Python callable -> ctypes -> this header".
 * Every entry point below names the reference interface whose arithmetic it
 * replaces.  All functions return 0 on success and a negative PSH_E* code on
 * failure; psh_last_error() then returns a thread-local description.
 *
 * Conventions
 *  - plain pointers and sizes only; images are row-major, x (columns) fastest.
 *  - "_dev" pointers are device pointers obtained from psh_malloc(); those
 *    calls are asynchronous on the library's HIP stream (psh_sync() to wait).
 *    "_host" entry points take host pointers, stage through device memory and
 *    return synchronously.
 *  - velocity is (2,m,n): plane 0 = u (px/step along x), plane 1 = v (along y),
 *    as in semilagrangian.py:44-46.
 *  - displacement is (2,m,n) float64 on both sides of the boundary, the dtype
 *    the reference returns (semilagrangian.py:201,264-266).
 */
#ifndef PYSTEPS_HIP_H
#define PYSTEPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSH_OK 0
#define PSH_EINVAL (-1)   /* bad argument (maps to ValueError in the Python shim) */
#define PSH_EHIP (-2)     /* HIP runtime error */
#define PSH_ENOTINIT (-3) /* psh_init() not called */
#define PSH_ENOMEM (-4)
#define PSH_ECOMM (-5)    /* RCCL error / librccl.so not loadable */
#define PSH_EUNSUPPORTED (-6)
#define PSH_EINPUT (-7)   /* input values the reference rejects (non-finite fields); see psh_semilag_host */

/* ---- runtime ----------------------------------------------------------- */
int psh_init(int device_id);          /* idempotent; binds the calling process to one GPU */
int psh_shutdown(void);
const char *psh_last_error(void);
const char *psh_version(void);
/* name_len bytes of name are filled (NUL-terminated); any pointer may be NULL */
int psh_device_info(int *device_id, int *cu_count, size_t *hbm_total, size_t *hbm_free,
                    char *name, int name_len);

/* knobs; "semilag_variant": 0 (default) the workgroup-window kernel - the motion field and the advected field of
 * a 64 x 32 tile's neighbourhood kept in LDS across lead steps - wherever it applies (interp_order 1 with a field,
 * n_iter >= 1, images >= 96 x 64, at least two sampling steps) and the gather kernels elsewhere;
 * 12 the window kernel for every eligible call; 7 gather kernels only: velocity from a packed {u,v} plane and the field
 * from a row-pair plane with dwordx4 loads (the default of rounds 2 - 4); 5 the same without the row-pair plane;
 * 1 one plane per component with DPP column sharing (what calls of fewer than 8 sampling steps take among the gather
 * kernels) - all of them bit-identical;
 * "idw_variant": 0 two-level pre-pass (64x64 supertile lists, one wave per 16x8 tile, two pixels per lane; default),
 * 1 one pre-pass per 16x16 tile;
 * "members_variant": members per thread of the member-batched step on packed planes: 2 (default: the two
 * trajectories' gathers overlap) or 1;
 * "lk_fused_nms": 1 = the resident dense Lucas-Kanade estimate computes the Shi-Tomasi response, the 3x3 maxima and the
 * candidate list in ONE pass without writing the response plane (lk_corner_response_nms), 0 (default) = response pass +
 * selection pass: same corners bit for bit; the fused pass measured slower;
 * "trim_cache": release the device blocks cached by psh_free (value ignored) */
int psh_set_option(const char *key, int value);

int psh_malloc(void **dev_ptr, size_t nbytes);
int psh_free(void *dev_ptr);
int psh_memcpy_h2d(void *dst_dev, const void *src_host, size_t nbytes); /* async; pageable src is staged */
int psh_memcpy_d2h(void *dst_host, const void *src_dev, size_t nbytes); /* synchronous */
/* queued on the library stream; dst should come from psh_host_alloc (pinned) - psh_sync() before it is read */
int psh_memcpy_d2h_async(void *dst_host, const void *src_dev, size_t nbytes);
int psh_memcpy_d2d(void *dst_dev, const void *src_dev, size_t nbytes);  /* async */
int psh_memset(void *dst_dev, int byte_value, size_t nbytes);           /* async */
int psh_sync(void);

/* HIP events on the library stream (bench.py times kernels with these) */
int psh_event_create(void **event);
int psh_event_destroy(void *event);
int psh_event_record(void *event);
int psh_event_elapsed_ms(void *start, void *stop, float *ms); /* waits for stop */

/* ---- element-wise passes either side of the path (keep a nowcast chain in HBM) ------ *
 * psh_db_transform_dev: pysteps/utils/transformation.py:150-232 (dB_transform).  Forward:
 *   out = R < threshold ? zerovalue : 10 log10(R)  (threshold in the units of R);  inverse:
 *   out = 10^(R/10) < 10^(threshold/10) ? zerovalue : 10^(R/10)  (threshold in dB).  n floats,
 *   in place allowed, NaN stays NaN.
 * psh_field_stats_dev: min / max over the finite values and the number of non-finite values
 *   (the NumPy scans of nowcasts/extrapolation.py:76 and semilagrangian.py:171-172). Synchronous. */
int psh_db_transform_dev(const float *in_dev, float *out_dev, size_t n, double threshold,
                         double zerovalue, int inverse);
/* float64 <-> float32 element conversion on the device (to_f64 != 0: float32 -> float64), so that
 * the float64 arrays pysteps works with cross the bus as they are instead of being narrowed /
 * widened by a host pass. */
int psh_convert_dev(const void *in_dev, void *out_dev, size_t n, int to_f64);
/* dst[i] += alpha * src[i] on float64 device arrays (queued on the library stream): base-position offsets of a custom
 * xy_coords grid added to / taken off a displacement field */
int psh_axpy_f64_dev(double *dst_dev, const double *src_dev, double alpha, size_t n);
/* pysteps/utils/check_norain.py:40-50 on a resident field: number of values > threshold (NaN never
 * counts) and np.nanmin of the field; a NaN threshold means "the minimum of the field"
 * (precip_thr=None).  Synchronous. */
int psh_count_above_dev(const float *in_dev, size_t n, double threshold, double *count_out, double *nanmin_out);
/* number of NaN / infinite values of a float64 device array (waits for the result) */
int psh_nonfinite_count_f64_dev(const double *in_dev, size_t n, double *count_out);
int psh_field_stats_dev(const float *in_dev, size_t n, double *min_out, double *max_out,
                        double *nonfinite_out);

/* ---- semi-Lagrangian extrapolation ------------------------------------- *
 * Replaces pysteps/extrapolation/semilagrangian.py:21-266 (extrapolate) incl.
 * its inner interpolate_motion (:181-198) and the scipy.ndimage.map_coordinates
 * order-0/1 resampling it calls (:185-190 mode="nearest", :225-232
 * mode="constant").  One fused kernel integrates every pixel's trajectory
 * through all T lead steps (n_iter midpoint sub-steps each) and writes one
 * advected plane per lead step.
 *
 *  precip      (m,n) float32 or NULL (displacement only; nowcasts/utils.py:498-503)
 *  velocity    (2,m,n) float32
 *  steps_host  T doubles, HOST memory: lead-time increments / vel_timestep
 *              (timestep_diff / vel_timestep of semilagrangian.py:165,198)
 *  n_iter      >= 0 (0 = no midpoint rule, :215-219)
 *  interp_order 0 .. 5 for the precip resampling (:85-90); 2 .. 5 = B-spline of that order incl. the
 *              spline prefilter and the two mask warps of :146-157,234-253 (outside -> NaN).
 *              The boundary mode of that resampling (map_coordinates_mode, :91-96,225-232) rides in
 *              the second byte: interp_order | PSH_MODE_* << 8 (outval is then the cval of
 *              "grid-constant"); with interp_order 3 the spline filter takes the mode's boundary
 *              condition and "nearest" / "grid-constant" are padded by 12 samples first, like SciPy.
 *  outval      value for pixels advected from outside the domain (may be NaN)
 *  disp        (2,m,n) float64 or NULL; resume = 1: it holds displacement_prev; resume = 2 (PSH_SL_RESUME_BASE): it
 *              holds the base positions of a custom xy_coords grid relative to the integer grid (see PSH_SL_BASE_IN_DISP)
 *              on entry (:203-207); if non-NULL it receives the final displacement
 *  out         (T,m,n) float32, required iff precip != NULL
 */
#define PSH_MODE_CONSTANT 0      /* scipy.ndimage mode names: "constant" (the default) */
#define PSH_MODE_NEAREST 1
#define PSH_MODE_REFLECT 2
#define PSH_MODE_MIRROR 3
#define PSH_MODE_WRAP 4
#define PSH_MODE_GRID_CONSTANT 5
#define PSH_MODE_GRID_WRAP 6
int psh_semilag_dev(const float *precip_dev, const float *velocity_dev, int m, int n,
                    const double *steps_host, int T, int n_iter, int interp_order,
                    float outval, double *disp_dev, int resume, float *out_dev);
/* The same with the motion field ALSO given as (m, n, 2) {u, v} pairs (velocity_uv_dev, 16-byte aligned; NULL: as
 * above) - the layout the kernel gathers from; without it every call interleaves the two planes first (0.04 ms at
 * 4096^2).  psh_dense_lk_uv_dev writes such a copy from its interpolation kernel.  The two arrays have to hold the
 * same field: the planes are still read where the pairs are not (other kernel variants, interp_order != 1). */
int psh_semilag_uv_dev(const float *precip_dev, const float *velocity_dev, const float *velocity_uv_dev, int m, int n,
                       const double *steps_host, int T, int n_iter, int interp_order, float outval, double *disp_dev,
                       int resume, float *out_dev);

/* 1 if the workgroup-window kernel (the default extrapolator for bilinear resampling of a field with n_iter >= 1;
 * it samples the velocity PLANES) takes (m, n) images, 0 if such images go to the gather kernels, which sample
 * {u, v} pairs - a caller that produces motion fields for them (psh_dense_lk_uv_dev) then writes the pair layout too
 * instead of leaving the interleaving to every extrapolation call.  Pure function: needs no device. */
int psh_semilag_window_shape(int m, int n);
/* Which kernel a call of this shape takes under the current "semilag_variant" (16-byte aligned planes assumed):
 * 12 the workgroup-window kernel (semilag_window), 7 / 5 / 1 the gather kernel (semilag_fused) on packed velocity +
 * row-pair field planes / packed velocity / one plane per component.  Pure function: needs no device. */
int psh_semilag_kernel(int m, int n, int T, int n_iter, int interp_order, int has_field);

/* Row-band form for output tiling across GPUs (BASELINE config 5: every rank holds the whole
 * input - it is tiny next to 288 GB - and advects only its band): pixels of rows
 * [row_begin, row_begin + row_count) are integrated and written to out (T,row_count,n);
 * disp, if given, stays full size (2,m,n) and only the band's rows are read/written. */
int psh_semilag_rows_dev(const float *precip_dev, const float *velocity_dev, int m, int n,
                         const double *steps_host, int T, int n_iter, int interp_order,
                         float outval, double *disp_dev, int resume, int row_begin, int row_count,
                         float *out_dev);

/* Host-buffer form (what the NumPy callers of the plugin reach): stages through device memory.
 * Transfer bound (4096^2 x 24: 1.4 ms of kernel, ~30 ms of PCIe): buffers allocated with
 * psh_host_alloc are copied directly at pinned-memory speed; any other host pointer is staged
 * through a ring of pinned chunks filled / drained by several host threads while the DMA engine
 * moves the previous chunk.  Thread-safe; the library mutex is released while the call waits.
 *
 * flags: PSH_SL_ALLOW_NONFINITE = allow_nonfinite_values (semilagrangian.py:106-137);
 *   PSH_SL_OUTVAL_MIN = outval "min": nanmin of precip (:171-172), the outval argument is ignored;
 *   PSH_SL_PRECIP_F64 / PSH_SL_VELOCITY_F64: the array holds float64 (narrowed on the device);
 *   PSH_SL_OUT_F64: out receives float64 (SciPy returns the dtype of its input).
 * The reference's input checks run as device reductions after the upload: *input_status (may be
 * NULL) receives PSH_SL_ST_* bits; if a condition the flags do not allow holds, nothing is computed
 * and PSH_EINPUT is returned (the shim raises the reference's ValueError). */
#define PSH_SL_ALLOW_NONFINITE 1
#define PSH_SL_OUTVAL_MIN 2
#define PSH_SL_PRECIP_F64 4
#define PSH_SL_VELOCITY_F64 8
#define PSH_SL_OUT_F64 16
/* disp_prev holds the BASE positions of a custom xy_coords grid relative to the integer grid (xy_coords - meshgrid,
 * semilagrangian.py:174-179), not a previous displacement: trajectories start there, the first increment is the
 * grid's own velocity (:203) and disp_out is relative to the integer grid too (the caller subtracts the offsets).
 * The *_dev entry points take the same through resume = 2 (PSH_SL_RESUME_BASE); resume = 1 is displacement_prev. */
#define PSH_SL_BASE_IN_DISP 32
#define PSH_SL_RESUME_BASE 2
#define PSH_SL_ST_PRECIP_NONFINITE 1
#define PSH_SL_ST_PRECIP_ALL_NONFINITE 2
#define PSH_SL_ST_VELOCITY_NONFINITE 4
#define PSH_SL_ST_VELOCITY_ALL_NONFINITE 8
int psh_semilag_host(const void *precip, const void *velocity, int m, int n,
                     const double *steps, int T, int n_iter, int interp_order, float outval,
                     const double *disp_prev, double *disp_out, void *out, int flags, int *input_status);

/* Pinned host blocks from a cached pool (limit: env PYSTEPS_HIP_PINNED_BYTES, default 16 GiB;
 * PSH_ENOMEM beyond it - callers then fall back to ordinary memory).  The Python shims build their
 * result arrays on these, so device-to-host copies land in the array the caller receives. */
int psh_host_alloc(void **host_ptr, size_t nbytes);
int psh_host_free(void *host_ptr);

/* Member-batched, stateful step for ensemble nowcasts: the worker of the generic nowcast
 * loop, pysteps/nowcasts/utils.py:441-462, for ALL members in one launch:
 *     velocity_j = V + par_j * V_par + perp_j * V_perp            (utils.py:448-451 with the BPS
 *                  perturbation of pysteps/noise/motion.py:146-180; V_par = V/|V|,
 *                  V_perp = (-V_par_y, V_par_x); par_j = g_par(t) eps_par_j / vsf, perp_j likewise)
 *     precip_j, D_j = extrapolate(precip_j, velocity_j, steps, displacement_prev=D_j)  (:453-458)
 *  precip (B,m,n) f32 or NULL; velocity (2,m,n); vhat = V_par (2,m,n) from
 *  psh_velocity_unit_dev, or NULL for "no perturbation"; pert_par/pert_perp: B doubles in HOST
 *  memory; disp (B,2,m,n) f64 in/out, stays resident between calls (resume=0: start from zero
 *  displacement); out (B,T,m,n) f32. */
int psh_velocity_unit_dev(const float *velocity_dev, int m, int n, float *vhat_dev);
int psh_semilag_members_dev(const float *precip_dev, const float *velocity_dev,
                            const float *vhat_dev, const double *pert_par_host,
                            const double *pert_perp_host, int n_members, int m, int n,
                            const double *steps_host, int T, int n_iter, int interp_order,
                            float outval, double *disp_dev, int resume, float *out_dev);
/* The same step with the trajectories kept in the kernel's own representation between calls:
 * state (B,m,n) records of 16 bytes {int32 P-x, int32 P-y, float32 frac_x, float32 frac_y} - half the
 * HBM traffic of the float64 displacement pair.  The converters translate to / from the
 * reference's displacement (B,2,m,n) float64 (D = (P - x) + frac). */
int psh_semilag_members_state_dev(const float *precip_dev, const float *velocity_dev,
                                  const float *vhat_dev, const double *pert_par_host,
                                  const double *pert_perp_host, int n_members, int m, int n,
                                  const double *steps_host, int T, int n_iter, int interp_order,
                                  float outval, void *state_dev, int resume, float *out_dev);
/* The same step gathering from an interleaved plane built once per motion field by
 * psh_members_pack_dev: (m,n,4) {u,v,V_par_x,V_par_y} when vhat is given, (m,n,2) {u,v} otherwise -
 * one dwordx4 per tap (or per tap row) instead of one dword per plane and tap: 4 loads per sampling
 * pass instead of 16.  Bit-identical results; velocity_dev / vhat_dev are still read on image borders. */
int psh_members_pack_dev(const float *velocity_dev, const float *vhat_dev, int m, int n, float *packed_dev);
int psh_semilag_members_packed_dev(const float *precip_dev, const float *velocity_dev,
                                   const float *vhat_dev, const float *packed_dev,
                                   const double *pert_par_host, const double *pert_perp_host,
                                   int n_members, int m, int n, const double *steps_host, int T,
                                   int n_iter, int interp_order, float outval, void *state_dev,
                                   int resume, float *out_dev);
int psh_members_state_to_disp_dev(const void *state_dev, int n_members, int m, int n, double *disp_dev);
int psh_members_disp_to_state_dev(const double *disp_dev, int n_members, int m, int n, void *state_dev);

/* ---- sparse vectors -> dense field: k-NN inverse distance weighting ------ *
 * Replaces pysteps/utils/interpolate.py:26-114 (idwinterp2d) as called from
 * pysteps/motion/lucaskanade.py:272-274, including the cKDTree k-NN query it
 * makes for every grid node (:80-86):
 *     w_i = (d_i / res + dist_offset)^-power over the k nearest samples,
 *     out = sum(w_i * values_i) / sum(w_i),  res = mean(|dx|,|dy|)  (:89-109).
 *  xy      (L,2) sample coordinates (x, y);  values (L,2) e.g. (u, v)
 *  grid    x_i = x0 + dx*i (i < n),  y_j = y0 + dy*j (j < m)   [np.arange -> 0,1]
 *  k       neighbours used (k >= L uses every sample; k <= 32 otherwise)
 *  out     (2,m,n): plane 0 = first value column, plane 1 = second
 * The _dev form takes float32 device arrays and reach_hint = an upper bound of
 * the distance between any grid node and any sample (sizes the search bins).
 * The _host form takes/returns float64 like the reference. */
int psh_idw_dev(const float *xy_dev, const float *values_dev, int L, int m, int n, double x0,
                double dx, double y0, double dy, int k, double power, double dist_offset,
                double reach_hint, float *out_dev);
int psh_idw_host(const double *xy, const double *values, int L, int m, int n, double x0,
                 double dx, double y0, double dy, int k, double power, double dist_offset,
                 double *out);

/* Radial-basis-function interpolant on a regular grid (pysteps/utils/interpolate.py:117-170 rbfinterp2d, which
 * wraps scipy.interpolate.Rbf): out(x) = sum_j weights_j phi(|x - xy_j|) for two variables at once.  The
 * weights come from the host (SciPy's own N x N solve); this is the N x m x n evaluation.  xy_dev (N,2) and
 * weights_dev (N,2) float64, out_dev (2,m,n) float64 at x = x0 + dx i, y = y0 + dy j; function: 0 multiquadric,
 * 1 inverse, 2 gaussian, 3 linear, 4 cubic, 5 quintic, 6 thin_plate; epsilon: Rbf's shape parameter.
 * Asynchronous on the library stream. */
int psh_rbf_eval_dev(const double *xy_dev, const double *weights_dev, int N, int m, int n, double x0, double dx,
                     double y0, double dy, int function, double epsilon, double *out_dev);

/* ---- scale-space blob detection: the `fd_method="blob"` feature detector (csrc/blob.hip) ---------------------- *
 * pysteps/feature/blob.py:32-140 -> scikit-image blob_log / blob_dog [third party] -> scipy.ndimage.gaussian_laplace /
 * gaussian_filter per scale, a 3 x 3 x 3 maximum_filter over the scale cube, peak mask.  SciPy's arithmetic operation
 * by operation (correlate1d's symmetric form in double, stored in the image's dtype after every pass, mode "reflect";
 * maximum_filter1d's ring of (value, death) pairs, which decides what the filter returns next to NaNs).
 *  psh_blob_cube_dev    image (m, n) float32 / float64 -> cube (K, m, n) float64:
 *      method 0 (LoG): cube[k] = -gaussian_laplace(image, sigma_k) * sigma_k^2,  K = nsig
 *      method 1 (DoG): cube[k] = (G(sigma_k) - G(sigma_k+1)) image * sigma_k,    K = nsig - 1
 *      radius_host[k] = int(4 sigma_k + 0.5); weights_host: per scale 2 (radius + 1) doubles - centre and distances
 *      1 .. radius of the smoothing kernel, then of the second-derivative kernel (scipy's _gaussian_kernel1d, evaluated
 *      by the caller with NumPy like SciPy does).  The kernels are queued; the host arrays may go when the call returns.
 *  psh_blob_peaks_dev   skimage peak_local_max(cube, threshold_abs, footprint ones(3,3,3), exclude_border False) as
 *      blob_log calls it: coords_host (capacity, 3) int32 (row, column, scale index) and their values, unordered;
 *      *count_host = number of peaks (above capacity: call again with more room).  Waits.
 *  psh_blob_gather_dev  values of one (m, n) float64 plane at `count` (row, column) pixels.  Waits. */
int psh_blob_cube_dev(const void *image_dev, int image_is_f32, int m, int n, int method, const double *sigmas_host, int nsig,
                      const int *radius_host, const double *weights_host, double *cube_dev);
int psh_blob_peaks_dev(const double *cube_dev, int K, int m, int n, double threshold, int capacity, int *coords_host,
                       double *values_host, int *count_host);
int psh_blob_gather_dev(const double *plane_dev, int m, int n, const int *yx_host, int count, double *values_host);

/* ---- dense Lucas-Kanade: image front end ----------------------------------- *
 * The NumPy + OpenCV stages of pysteps/motion/lucaskanade.py:205-242, per frame /
 * frame pair.  OpenCV is a third-party dependency of the reference (not in its
 * tree); the kernels follow the OpenCV 4.x algorithms restated in
 * oracle/lk_opencv.py.
 *
 * psh_lk_prepare_dev  - one frame (m,n) float32 (NaN/Inf = missing):
 *     fill + binary opening with the 3x3 cross (pysteps/utils/images.py:58-86,
 *     cv2.morphologyEx OPEN) -> clean (m,n) float32 (missing pixels stay NaN);
 *     min-max rescale to uint8 by truncation for the tracker
 *     (pysteps/tracking/lucaskanade.py:135-160) -> track_u8, and for the feature
 *     detector (pysteps/feature/shitomasi.py:128-151 incl. the row-0/1 masking of
 *     :140) -> feature_u8 (may be NULL).  stats_dev: 8 floats of device scratch that
 *     carry min/max/NaN-count between the kernels (no host round trip).
 * psh_lk_corners_dev  - cv2.goodFeaturesToTrack(maxCorners, qualityLevel,
 *     minDistance, blockSize, useHarris=False, mask) of shitomasi.py:153-165 with
 *     mask = NOT dilate(missing, ones(buffer_mask)) (:135-139,152).  Writes up to
 *     max_corners (x,y) float32 pairs to HOST memory, strongest first.
 * psh_lk_track_dev    - cv2.calcOpticalFlowPyrLK(prev, next, points, winSize,
 *     maxLevel, criteria=(COUNT+EPS, max_count, epsilon), minEigThreshold) of
 *     tracking/lucaskanade.py:164-171: builds both Gaussian pyramids and the
 *     Scharr gradients on device, tracks every point in one launch.  points /
 *     next_points / status are HOST arrays (p,2) float32, (p,2) float32, (p) uint8.
 * All three are ordered on the library stream; the last two return synchronously. */
int psh_lk_prepare_dev(const float *frame_dev, int m, int n, int size_opening, int buffer_mask,
                       float *clean_dev, unsigned char *track_u8_dev,
                       unsigned char *feature_u8_dev, float *stats_dev);
/* psh_lk_prepare_dev for a float64 frame: minimum, opening test, rescale and truncation in double, as
 * the reference computes them for float64 input; clean_dev receives the float32 copy of the cleaned frame. */
int psh_lk_prepare_f64_dev(const double *frame_dev, int m, int n, int size_opening, int buffer_mask,
                           float *clean_dev, unsigned char *track_u8_dev,
                           unsigned char *feature_u8_dev, float *stats_dev);
int psh_lk_corners_dev(const unsigned char *feature_u8_dev, const float *clean_dev,
                       float *stats_dev, int m, int n, int block_size, int buffer_mask,
                       double quality_level, double min_distance, int max_corners,
                       float *points_host, int *count_host);
int psh_lk_track_dev(const unsigned char *prev_u8_dev, const unsigned char *next_u8_dev, int m,
                     int n, const float *points_host, int npts, int win_w, int win_h,
                     int max_level, int max_count, double epsilon, double min_eig_threshold,
                     float *next_points_host, unsigned char *status_host);
/* The same two operations in halves, so that independent device work overlaps the host's
 * ordered pass over the corner candidates: corners_launch queues the kernels and the copy of
 * the candidates, pyramids builds the Gaussian pyramids of a frame pair (plus a Scharr gradient
 * image for windows wider than 61 columns; opaque handle, release with psh_lk_pyramids_free),
 * corners_finish waits only for the candidates, track_pyr tracks points through a prebuilt
 * pyramid set.  Up to four corner requests may be in flight; corners_finish answers them first
 * in, first out. */
/* The ordered min-distance pass of goodFeaturesToTrack alone (featureselect.cpp; pure host
 * code): keys = candidates in walking order, (response bits << 32) | (y * n + x), strongest first,
 * ties by higher address first; accepted corners -> points_host (x, y) float32, at most max_corners. */
int psh_lk_greedy_host(const unsigned long long *keys, int count, int m, int n, double min_distance,
                       int max_corners, float *points_host, int *count_host);
/* The same pass on the DEVICE (csrc/lk_sparse.hip corner_order, the kernel the corner entry points
 * and psh_dense_lk_dev run): keys in ANY order (host array), every response in
 * (response_max * quality_level, response_max]; the head of the descending order is selected by a
 * key histogram, ordered in LDS and walked in batches.  Same result as sorting the keys descending
 * and calling psh_lk_greedy_host.  PSH_EUNSUPPORTED beyond 2048 corners / 65535 rows or columns. */
int psh_lk_order_host(const unsigned long long *keys_host, int count, float response_max,
                      double quality_level, int m, int n, double min_distance, int max_corners,
                      float *points_host, int *count_host);
int psh_lk_corners_launch_dev(const unsigned char *feature_u8_dev, const float *clean_dev,
                              float *stats_dev, int m, int n, int block_size, int buffer_mask,
                              double quality_level, double min_distance, int max_corners);
int psh_lk_corners_finish(float *points_host, int *count_host);
int psh_lk_pyramids_dev(const unsigned char *prev_u8_dev, const unsigned char *next_u8_dev, int m,
                        int n, int win_w, int win_h, int max_level, void **handle_out);
int psh_lk_pyramids_free(void *handle);

/* Row bands of the image passes for frames tiled over several GPUs (BASELINE config 5).  Every rank
 * holds the whole frame and processes rows [e0, e1) - its own rows [r0, r1) plus a halo - as a
 * sub-image; results land at their absolute rows in full-size buffers and are exact except within a
 * few rows of band edges that are not frame borders.  Statistics and corner candidates are taken
 * from the OWN rows only and combined across ranks by the caller:
 *   stats   -> stats[0] min, stats[1] NaN count of the own rows             then allreduce MIN / SUM
 *   open    -> opening of the band (needs the global stats[0..1]); stats[2] max, stats[3] / stats[4]
 *              min / max of the feature rendering over the own rows          then allreduce MAX / MIN / MAX
 *   to_u8   -> both uint8 renderings of the band (needs the global stats[0..4])
 *   response-> Shi-Tomasi response of the band, stats[5] its maximum over the own rows   then allreduce MAX
 *   select  -> candidates of the own rows as (response bits << 32 | absolute pixel address) keys, any
 *              order (the caller gathers the keys of all ranks, sorts them descending and runs
 *              psh_lk_greedy_host - the walking order of goodFeaturesToTrack)
 * psh_lk_pyramids_band turns a pyramid built by psh_lk_pyramids_dev from the band sub-images (rows
 * [band_first_row, ...) of a frame_rows-row frame; band_first_row a multiple of 2^levels) into a view of
 * the whole-frame pyramid: points are given and tracked in whole-frame coordinates (bit-identical
 * arithmetic), only the stored rows differ.  Tracks whose windows leave the rows that equal the
 * whole-frame pyramid come back with status bit 1 (value 2 / 3) set and have to be redone on
 * whole-frame data.  PSH_EUNSUPPORTED if the band is too small for the frame's number of levels. */
int psh_lk_band_stats_dev(const float *frame_dev, int m, int n, int r0, int r1, float *stats_dev);
int psh_lk_band_open_dev(const float *frame_dev, int m, int n, int e0, int e1, int r0, int r1, int size_opening,
                         int buffer_mask, float *clean_dev, float *stats_dev);
int psh_lk_band_to_u8_dev(const float *clean_dev, int m, int n, int e0, int e1, int buffer_mask, const float *stats_dev,
                          unsigned char *track_u8_dev, unsigned char *feature_u8_dev);
int psh_lk_band_response_dev(const unsigned char *feature_u8_dev, const float *clean_dev, int m, int n, int e0, int e1,
                             int r0, int r1, int block_size, int buffer_mask, float *stats_dev, float *eig_dev);
int psh_lk_band_select_dev(const float *eig_dev, const float *clean_dev, int m, int n, int e0, int e1, int r0, int r1,
                           int buffer_mask, double quality_level, const float *stats_dev,
                           unsigned long long *keys_dev, int cap, int *count_dev);
int psh_lk_pyramids_band(void *handle, int frame_rows, int band_first_row, int *top_level_out);
int psh_lk_track_pyr_dev(void *handle, const float *points_host, int npts, int max_count,
                         double epsilon, double min_eig_threshold, float *next_points_host,
                         unsigned char *status_host);

/* Whole dense_lucaskanade (pysteps/motion/lucaskanade.py:182-279, default detector and
 * interpolator) in one call.  With field_dev the estimate is ONE chain of kernel launches on the
 * library stream (corner ordering, tracking, pooling, outlier test, declustering and IDW all read
 * their counts from device memory): the call returns without waiting for the device unless
 * count_out is given (then it waits for the 4-byte sample count).
 *  frames (nframes,m,n) f32 device (NaN/Inf = missing).  field_dev (2,m,n) f32 or NULL; with
 *  NULL the sparse vectors after outlier removal are returned instead (dense=False):
 *  xy_host / uv_host (capacity,2) f64, *count_out rows.  Fields of psh_lk_params follow the
 *  reference's keyword arguments (lk_kwargs / fd_kwargs / interp_kwargs and the function's own). */
typedef struct psh_lk_params {
  int size_opening;          /* 0 or 3 (lucaskanade.py:48, images.py:27) */
  int buffer_mask;           /* shitomasi.py:33 */
  int max_corners;           /* max_corners / max_num_features */
  int block_size;            /* odd, <= 7 */
  double quality_level, min_distance;
  int win_w, win_h;          /* winsize */
  int max_level;             /* nr_levels */
  int max_count;             /* criteria: iteration count */
  double epsilon;            /* criteria: epsilon */
  double min_eig_threshold;  /* min_eig_thr */
  double nr_std_outlier;
  int k_outlier;
  double decl_scale;
  int idw_k;                 /* <= 0: use every vector (k=None) */
  double idw_power, idw_dist_offset;
  int frames_f64;            /* != 0: frames_dev points to float64 planes (psh_lk_prepare_f64_dev) */
} psh_lk_params;
int psh_dense_lk_dev(const float *frames_dev, int nframes, int m, int n, const psh_lk_params *params,
                     float *field_dev, double *xy_host, double *uv_host, int capacity,
                     int *count_out);
/* ... with the dense field once more as (m, n, 2) {u, v} pairs in field_uv_dev (may be NULL; needs field_dev) */
int psh_dense_lk_uv_dev(const float *frames_dev, int nframes, int m, int n, const psh_lk_params *params,
                        float *field_dev, float *field_uv_dev, double *xy_host, double *uv_host, int capacity,
                        int *count_out);

/* ---- sparse vector QC: local Mahalanobis outlier test ----------------------- *
 * The form of pysteps/utils/cleansing.py:124-249 (detect_outliers) used by dense LK
 * (pysteps/motion/lucaskanade.py:252): for every 2-vector the k nearest OTHER
 * samples by coordinate (the reference: cKDTree k+1 query minus the first hit),
 * Mahalanobis distance to their mean with their covariance (ddof=1); flag = 1 iff
 * > thr; singular covariance -> 0.  HOST arrays: xy (n,2) f64, values (n,2) f64,
 * flags (n) uint8.  float64 on device; equidistant neighbours: lower index first. */
int psh_outliers_local_host(const double *xy, const double *values, int n, int k, double thr,
                            unsigned char *flags);

/* decluster of pysteps/utils/cleansing.py:21-121 for (n,2) coordinates, (n,2) values and a
 * scalar scale: per scale-sized cell (lexicographic cell order) the component-wise medians.
 * Pure host code (no device needed).  out_xy / out_values must hold n rows; *out_count rows
 * are written. */
int psh_decluster_host(const double *xy, const double *values, int n, double scale,
                       int min_samples, double *out_xy, double *out_values, int *out_count);
/* What psh_dense_lk_dev does between the outlier test and the interpolation, as ONE device kernel
 * (csrc/lk_sparse.hip vectors_finish; pysteps/motion/lucaskanade.py:254-274 +
 * pysteps/decorators.py:199-208): drop the flagged vectors (none if count < 2), decluster with
 * min_samples 1 when decl_scale > 1 (coordinates inside a 65535 x 65535 image), then the
 * interpolator preamble: *out_mode 1 = constant field out_const[0..1] (no vector -> zeros, one
 * vector, all values equal), 0 = interpolate the *out_count float32 samples out_xy / out_values
 * (count rows each); *out_reach = farthest a sample can be from a node of the (m,n) pixel grid.
 * HOST arrays in and out; at most 8192 vectors. */
int psh_vectors_finish_host(const double *xy, const double *values, const unsigned char *outlier_flags,
                            int count, double decl_scale, int m, int n, float *out_xy, float *out_values,
                            int *out_count, int *out_mode, float *out_const, float *out_reach);

/* ---- two-dimensional FFTs, float64 / complex128 (csrc/fft.hip) ----------------- *
 * What the FFT method object of pysteps provides (pysteps/utils/fft.py:20-37: numpy.fft.rfft2,
 * irfft2(X, s=shape), fft2, ifft2) and the STEPS member loop calls ~10 times per member and lead
 * time (pysteps/noise/fftgenerators.py:330-400, pysteps/cascade/decomposition.py:77-262,
 * pysteps/nowcasts/steps.py:1111,1189).  numpy's conventions: no scaling forward, 1/(m n) backward,
 * rfft2 keeps the n/2+1 non-negative frequencies of the last axis, irfft2 ignores the imaginary
 * parts of its zero and Nyquist bins.  Row-major device arrays; each side a power of two in 2..8192
 * or any other length in 2..4096 (Bluestein's chirp-z identity inside the same kernels;
 * PSH_EUNSUPPORTED beyond: the Python shim hands those to numpy.fft).  Asynchronous on the
 * library stream.  Complex numbers are (re, im) float64 pairs.
 *  psh_fft_rfft2_dev   in (m,n) f64           -> out (m,n/2+1) c128
 *  psh_fft_irfft2_dev  in (m,n/2+1) c128      -> out (m,n) f64        (in is left untouched)
 *  psh_fft_c2c2_dev    in (m,n) c128          -> out (m,n) c128, inverse != 0: ifft2 (in == out allowed) */
int psh_fft_rfft2_dev(const double *in_dev, int m, int n, void *out_dev);
int psh_fft_irfft2_dev(const void *in_dev, int m, int n, double *out_dev);
/* ... with np.min of the result as the order-preserving key psh_steps_mask_dev reads (psh_field_min_key_dev's
 * result without its sweep over the field): the row pass of the transform folds its outputs in */
int psh_fft_irfft2_min_dev(const void *in_dev, int m, int n, double *out_dev, unsigned long long *min_key_dev);
int psh_fft_c2c2_dev(const void *in_dev, int m, int n, int inverse, void *out_dev);

/* ---- spectral building blocks of the STEPS member loop (csrc/cascade.hip) -------- *
 * float64 device arrays, sizes as for the FFTs.
 *  psh_cascade_decompose_dev  pysteps/cascade/decomposition.py:77-262 (decomposition_fft), spatial in /
 *      spatial out, no mask: levels[k] = irfft2(rfft2(field [- mean]) * weights[k]), k < nlevels;
 *      weights (nlevels, m, n/2+1) = bp_filter["weights_2d"]; means / stds (np.mean, np.std) of the
 *      levels come back in HOST arrays of nlevels doubles (the call waits for them; both NULL: nothing
 *      is returned and the call is asynchronous - the resident member loop); normalize != 0:
 *      levels[k] = (levels[k] - mean_k) / std_k; subtract_mean != 0: *field_mean_host = mean(field).
 *  psh_cascade_recompose_dev  decomposition.py:265-305 (recompose_fft): out = sum_k levels[k] * stds[k]
 *      + means[k] (plain sum if means_host == stds_host == NULL) + field_mean.  Asynchronous.
 *  psh_noise_filter_dev       pysteps/noise/fftgenerators.py:420-433: out = irfft2(rfft2(white) * filter),
 *      standardised to zero mean and unit (population) variance; filter (m, n/2+1).  Asynchronous. */
int psh_cascade_decompose_dev(const double *field_dev, const double *weights_dev, int nlevels, int m, int n,
                              int normalize, int subtract_mean, double *levels_dev, double *means_host,
                              double *stds_host, double *field_mean_host);
/* The same with the levels left as the transforms produce them and (mean, std) per level written to
 * stats_dev (nlevels pairs of doubles, device): nothing waits on the host; the standardisation of
 * decomposition.py:224-232 is applied by the consumer, psh_steps_ar_recompose_raw_dev, on the way in. */
int psh_cascade_decompose_stats_dev(const double *field_dev, const double *weights_dev, int nlevels, int m, int n,
                                    double *levels_dev, double *stats_dev);
int psh_cascade_recompose_dev(const double *levels_dev, int nlevels, int m, int n, const double *means_host,
                              const double *stds_host, double field_mean, double *out_dev);
int psh_noise_filter_dev(const double *white_dev, const double *filter_dev, int m, int n, double *out_dev);
/*  psh_ar_iterate_dev         pysteps/timeseries/autoregression.py:1020-1070 (iterate_ar_model): x (nt, plane),
 *      phi (p + 1 host doubles, p in 1..8, nt >= p), eps (plane) or NULL; out (nt, plane) = [x[1], ..., x[nt-1],
 *      phi[0] x[nt-1] + phi[1] x[nt-2] + ... + phi[p] eps], products rounded before the additions like the NumPy
 *      expression (bit-identical).  out must not overlap x.  Asynchronous. */
int psh_ar_iterate_dev(const double *x_dev, int nt, size_t plane, const double *phi_host, int p,
                       const double *eps_dev, double *out_dev);

/* ---- empirical-CDF probability matching of the member loops (csrc/probmatch.hip) -------- *
 *  psh_probmatch_dev  pysteps/postprocessing/probmatching.py:55-140, nonparam_match_empirical_cdf(initial,
 *      target) with ignore_indices=None (nowcasts/steps.py:1199, sprog.py:421, sseps.py:783,804): out[i] =
 *      sorted(target')[rank of initial[i]], pixels at the minimum of initial -> minimum of target; target' =
 *      target with its NaNs at the minimum and, if it has more wet pixels than initial, everything below
 *      np.percentile(target, 100 * (1 - wet fraction of initial)) at the minimum.  count float64 values
 *      each, out must not overlap the inputs.  Tied wet values of initial are ranked in pixel order.
 *      The call waits for the result's status: PSH_EINVAL with the reference's messages (initial all NaN /
 *      not finite), PSH_EUNSUPPORTED for inputs left to the reference (more than 16384 wet values tied or
 *      in one of the 2^20 value buckets, infinities or no finite value in target). */
int psh_probmatch_dev(const double *initial_dev, const double *target_dev, size_t count, double *out_dev);
/* the same without the wait: *status_dev (device int) receives the outcome, to be read by the caller later
 * and turned into the return code / error text of psh_probmatch_dev by psh_probmatch_status() - the
 * resident member loop queues every member's matching and waits once per time step */
int psh_probmatch_async_dev(const double *initial_dev, const double *target_dev, size_t count, double *out_dev,
                            int *status_dev);
int psh_probmatch_status(int status);
/* A target that does not change between calls - the observation nowcasts/steps.py:1199 matches every member
 * against at every time step; the reference argsorts it again each time (probmatching.py:120-126) - needs its
 * half of the work once: a plan keeps the target's statistics, verdict and sorted wet values.
 * psh_probmatch_planned_dev(plan, initial, count, out, status_dev) = psh_probmatch_async_dev against the plan's
 * target (status_dev NULL: waits and returns the verdict like psh_probmatch_dev).  Identical outputs. */
/* k-th smallest value (0-based) of a NaN-free field: what compute_percentile_mask (pysteps/nowcasts/utils.py:102-138)
 * takes out of its full sort for the S-PROG precipitation mask (nowcasts/steps.py:1113-1114). Waits. */
int psh_order_statistic_dev(const double *field_dev, size_t count, size_t index, double *value_host);
int psh_probmatch_plan_create(const double *target_dev, size_t count, void **plan_out);
int psh_probmatch_plan_destroy(void *plan);
int psh_probmatch_planned_dev(const void *plan, const double *initial_dev, size_t count, double *out_dev, int *status_dev);
/* psh_steps_mask_dev(field, count, grey, keep, min_key) followed by psh_probmatch_planned_dev(plan, field, count, out,
 * status_dev) (nowcasts/steps.py:1221-1240, then :1198-1201) with the mask applied by the matching's own first sweep over
 * the field (the one that takes its statistics): one pass less, identical results; field_dev holds the masked field
 * afterwards; out_dev must be another array */
int psh_steps_mask_probmatch_dev(const void *plan, double *field_dev, size_t count, const double *grey_mask_dev,
                                 const unsigned char *keep_mask_dev, const unsigned long long *min_key_dev,
                                 double *out_dev, int *status_dev);

/* ---- incremental precipitation mask of the member loops (csrc/mask.hip) -------- *
 *  psh_dilated_mask_dev  pysteps/nowcasts/utils.py:69-101, compute_dilated_mask(input_mask, kr, r) (nowcasts/
 *      steps.py:983,1210, sseps.py:472,821): mask (m,n) uint8 on the device, non-zero = set; kr (kh,kw) uint8
 *      on the HOST, the structuring element of the first dilation (origin at its centre, at most 1024 set
 *      elements); r in 0..254 rim iterations with the 4-neighbour cross; out (m,n) float64 = (mask0 + sum_k
 *      dilate^k(mask0)) / its maximum, evaluated as max(0, r + 1 - L1 distance to mask0) / (r + 1) - bit-identical
 *      with the reference (NaN everywhere if nothing is set, like 0 / 0 in NumPy).  Asynchronous. */
int psh_dilated_mask_dev(const unsigned char *mask_dev, int m, int n, const unsigned char *kr_host, int kh, int kw,
                         int r, double *out_dev);
/* `field >= threshold` (nowcasts/steps.py:1211) and psh_dilated_mask_dev of the result in one entry point, on bit
 * masks: one pass over the float64 field (m, n) -> one bit per pixel, then one kernel for the structure's dilation,
 * the r cross dilations and the float64 mask (a wave per tile, rows as lanes, columns as bits of a 64-bit word).
 * Bit-identical with psh_ge_mask_dev + psh_dilated_mask_dev.  PSH_EUNSUPPORTED (take those two): a structure
 * without its centre element, r + the structure's reach above 24 pixels. */
int psh_steps_incremental_mask_dev(const double *field_dev, int m, int n, double threshold,
                                   const unsigned char *kr_host, int kh, int kw, int r, double *out_dev);

/* ---- element-wise half of one STEPS member update (csrc/steps_loop.hip) ------------ *
 * pysteps/nowcasts/steps.py:1057-1219 `__update_state` between the spectral operators, the CDF
 * matching and the incremental mask; float64, every product and sum rounded on its own like the
 * NumPy expressions (bit-identical arithmetic).  All asynchronous on the library stream.
 *  psh_steps_ar_recompose_dev  all cascade levels of ONE member: cascades (nlevels, p, plane) holds the
 *      AR history of every level as a ring - x[-1-j] in slot (head + p - 1 - j) % p; the new value
 *      x_new = 0.0 + sum_j phi[k][j] x[-1-j] + phi[k][p] (eps[k] * eps_scale[k])   (steps.py:1131-1140,
 *      timeseries/autoregression.py:1056-1070) overwrites slot `head` (the caller advances head by one,
 *      modulo p); field = sum_k (x_new[k] * sigma[k] + mu[k])   (cascade/decomposition.py:294-301).
 *      phi_host (nlevels, p+1); eps_dev (nlevels, plane) or NULL; min_key_dev (may be NULL): np.min(field)
 *      as an order-preserving 64-bit key in device memory, consumed by psh_steps_mask_dev.
 *      nlevels <= 16, p <= 8 (PSH_EUNSUPPORTED beyond).
 *  psh_steps_mask_dev  steps.py:1221-1240: grey_mask_dev (float64, mask_method="incremental"):
 *      field = min + (field - min) * mask, then field = min wherever not field > min;
 *      keep_mask_dev (uint8, "obs" / "sprog"): field = min where the mask is 0.  Exactly one of the two.
 *  psh_steps_mean_shift_dev  steps.py:1203-1206 (probmatching_method="mean"): values >= threshold get
 *      v - mean(those values) + mu_0 (the mean by a fixed tree of partial sums: ~1e-16 from np.mean)
 *  psh_ge_mask_dev     out[i] = field[i] >= threshold (uint8; NaN -> 0)      (steps.py:1211)
 *  psh_nan_where_dev   field[i] = NaN where mask[i] != 0                      (steps.py:1217)
 *  psh_lerp_dev        out = (1 - w) * a + w * b                              (nowcasts/utils.py:419-427) */
int psh_steps_ar_recompose_dev(double *cascades_dev, int nlevels, int p, size_t plane, int head,
                               const double *phi_host, const double *eps_dev, const double *eps_scale_host,
                               const double *mu_host, const double *sigma_host, double *field_dev,
                               unsigned long long *min_key_dev);
/* eps_dev = unnormalised noise levels, eps_stats_dev = their (mean, std) pairs from
 * psh_cascade_decompose_stats_dev: eps_k = (eps_k - mean_k) / std_k first, then as above (bit-identical
 * with standardising in a pass of its own). */
int psh_steps_ar_recompose_raw_dev(double *cascades_dev, int nlevels, int p, size_t plane, int head,
                                   const double *phi_host, const double *eps_dev, const double *eps_stats_dev,
                                   const double *eps_scale_host, const double *mu_host, const double *sigma_host,
                                   double *field_dev, unsigned long long *min_key_dev);
/* The same member update with the AR history kept as SPECTRA (rfft2 of the level fields): everything between the white
 * noise and the recomposed field is linear except two standardisations, whose second moments Parseval's identity reads
 * off the spectrum - two transforms per member update instead of nine (what the reference's domain="spectral" option
 * does, nowcasts/steps.py:122-126; same numbers as the spatial chain up to rounding).
 *   psh_steps_spectral_sums_dev  noise_spec = rfft2(white) (m, n/2+1) complex128, filter (m, n/2+1), weights
 *                                (nlevels, m, n/2+1) float64 -> sums[k] = sum' |noise filter weights_k|^2 (device)
 *   psh_steps_spectral_ar_dev    cascades (nlevels, p, m, n/2+1) complex128 rings (slot `head` = oldest, overwritten):
 *                                X_k <- sum_j phi_kj X_k[-1-j] + phi_kp noise_std_k m n / sqrt(sums_k) noise filter weights_k;
 *                                field_spec = sum_k sigma_k X_k + (sum_k mu_k) m n at DC   (field = irfft2(field_spec))
 *   psh_field_min_key_dev        np.min of a field as the order-preserving key psh_steps_mask_dev reads */
int psh_steps_spectral_sums_dev(const void *noise_spec_dev, const double *filter_dev, const double *weights_dev, int nlevels,
                                int m, int n, double *sums_dev);
int psh_steps_spectral_ar_dev(void *cascades_dev, int nlevels, int p, int m, int n, int head, const double *phi_host,
                              const void *noise_spec_dev, const double *filter_dev, const double *weights_dev,
                              const double *sums_dev, const double *noise_std_host, const double *mu_host,
                              const double *sigma_host, void *field_spec_dev);
/* The reference's OWN domain="spectral" member update (nowcasts/steps.py:1116-1166 with params["domain"] == "spectral";
 * noise/fftgenerators.py:407-437, cascade/decomposition.py:195-262 with spectral input and output, recompose_fft:284-300):
 * the state is spectral there (per level the coefficients where the band-pass weight exceeds 1e-12), the noise a field
 * of unit phasors exp(i theta), theta (m, n/2+1) float64 from RandomState.uniform(0, 2 pi) (psh_rng_uniform_dev).
 *   cascades (nlevels, p, m, n/2+1) complex128 rings of FULL half-spectrum planes (slot `head` = oldest, overwritten;
 *   entries outside a level's mask are not touched);  inv_std_noise = 1 / spectral.std(F with F[0,0] = 0),
 *   inv_std_levels[k] = 1 / spectral.std(F / std_noise * W_k) - constants of a nowcast since |exp(i theta)| = 1;
 *   field_spec = sum_k mask_k (sigma_k X_k + mu_k)   (field = irfft2(field_spec));
 *   theta_dev == NULL: no innovation term (the deterministic model behind the S-PROG mask, steps.py:1089-1114) */
int psh_steps_phase_ar_dev(void *cascades_dev, int nlevels, int p, int m, int n, int head, const double *phi_host,
                           const double *theta_dev, const double *filter_dev, const double *weights_dev,
                           double inv_std_noise, const double *inv_std_levels_host, const double *noise_std_host,
                           const double *mu_host, const double *sigma_host, void *field_spec_dev);
/* level planes from the reference's compact spectral arrays (cascade/decomposition.py:233-236: field[weights > 1e-12],
 * row-major): psh_mask_row_offsets_dev -> offsets (m + 1) int32 of one level's weights plane (m <= 8192);
 * psh_expand_compact_c128_dev -> dst (m, nc) complex128 = src scattered to the kept coefficients, zero elsewhere */
int psh_mask_row_offsets_dev(const double *weights_dev, int m, int nc, int *offsets_dev);
int psh_expand_compact_c128_dev(const double *weights_dev, int m, int nc, const int *offsets_dev, const void *src_dev,
                                void *dst_dev);
int psh_field_min_key_dev(const double *field_dev, size_t n, unsigned long long *min_key_dev);
int psh_steps_mask_dev(double *field_dev, size_t n, const double *grey_mask_dev, const unsigned char *keep_mask_dev,
                       const unsigned long long *min_key_dev);
int psh_steps_mean_shift_dev(double *field_dev, size_t n, double threshold, double mu_0);
int psh_ge_mask_dev(const double *field_dev, size_t n, double threshold, unsigned char *out_dev);
int psh_nan_where_dev(double *field_dev, const unsigned char *mask_dev, size_t n);
int psh_lerp_dev(const double *a_dev, const double *b_dev, double w, double *out_dev, size_t n);

/* ---- numpy.random.RandomState.randn on the device (csrc/rng.hip) ------------------ *
 * The white noise of the STEPS member loop: pysteps/noise/fftgenerators.py:400 draws
 * `randstate.randn(m, n)` per member and time step from generators seeded by the chain of
 * pysteps/nowcasts/steps.py:885-898.  NumPy's legacy generator (third party: MT19937 + polar method,
 * numpy/random/src/legacy/legacy-distributions.c `legacy_gauss`) restated for n_streams independent
 * generators; words, positions, accept / reject decisions and the final state are bit-identical with
 * NumPy, the values wherever the C library's log() is correctly rounded (1 ulp otherwise, csrc/cr_log.h).
 *  psh_rng_create     keys (n_streams, 624) uint32, pos / has_gauss / gauss per stream: the fields of
 *                     RandomState.get_state() (has_gauss_host / gauss_host may be NULL); max_draw = the
 *                     largest `count` psh_rng_randn_dev will be asked for; n_draws_hint > 0: that many draws
 *                     are expected - the streams are cut into chunks of 512 blocks whose start states are
 *                     computed by jump-ahead (x^J mod the generator's characteristic polynomial, tables from
 *                     tools/gen_mt_jump.py) so that every chunk is produced by its own workgroup; 0: one
 *                     workgroup per stream produces the words in sequence (also what happens beyond the hint)
 *  psh_rng_randn_dev  out (n_streams, count) float64: the next `count` values of every stream, like
 *                     `randn(count)`.  Asynchronous; side != 0 runs the draw on the handle's own stream
 *                     behind everything queued on the library stream so far (the noise of the next time
 *                     step beside the member loop): psh_rng_wait() makes the library stream wait for it
 *                     and has to be called before out_dev is read or freed
 *  psh_rng_uniform_dev  out (n_streams, count) float64: the next `count` values of every stream, like
 *                     `uniform(low, high, count)` (low + (high - low) * random_sample(): two words per value);
 *                     same stream rules as psh_rng_randn_dev.  What the spectral-domain noise generator draws
 *                     (pysteps/noise/fftgenerators.py:407)
 *  psh_rng_check      PSH_EHIP when a draw that has COMPLETED found fewer accepted attempts in its window than
 *                     it needed (a > 10 sigma event; the tail of that draw is unwritten); reads one pinned host
 *                     word, no device work - meant to be called once per time step after a wait the caller has
 *                     anyway.  The flag stays raised; get_state reports the same condition
 *  psh_rng_get_state  waits for the draws and returns the generators' states in get_state() form
 *                     (RandomState.set_state() then continues the stream on the host) */
int psh_rng_create(int n_streams, const uint32_t *keys_host, const int *pos_host, const int *has_gauss_host,
                   const double *gauss_host, size_t max_draw, int n_draws_hint, void **handle_out);
int psh_rng_randn_dev(void *handle, size_t count, double *out_dev, int side);
int psh_rng_uniform_dev(void *handle, size_t count, double low, double high, double *out_dev, int side);
int psh_rng_wait(void *handle);
int psh_rng_check(void *handle);
int psh_rng_get_state(void *handle, uint32_t *keys_host, int *pos_host, int *has_gauss_host, double *gauss_host);
int psh_rng_destroy(void *handle);

/* ---- multi-GPU: RCCL over xGMI, one rank (process) per GPU ------------------ *
 * The reference has no communication layer (single process, optional dask threads:
 * pysteps/nowcasts/utils.py:464-471); members / fields shard across GPUs with ONE
 * broadcast of the input fields.  librccl.so is dlopen'ed on first use.
 *  psh_comm_unique_id   rank 0: 128-byte ncclUniqueId to hand to every rank (any host channel)
 *  psh_comm_init        collective: ncclCommInitRank on the GPU bound by psh_init()
 *  psh_comm_broadcast   in-place ncclBroadcast of nbytes from root, on the library stream
 *  psh_comm_allgather   ncclAllGather of nbytes_per_rank (sparse vector lists), library stream
 *  psh_comm_allreduce_f32  in-place ncclAllReduce of `count` floats (PSH_COMM_MIN / MAX / SUM): the global
 *                       min / max of the uint8 rescales and the maximum corner response of the row-band
 *                       Lucas-Kanade passes (tracking/lucaskanade.py:143-160, shitomasi.py:143-151) */
#define PSH_COMM_SUM 0
#define PSH_COMM_MAX 1
#define PSH_COMM_MIN 2
int psh_comm_unique_id_bytes(void);
int psh_comm_unique_id(void *id_out);
int psh_comm_init(const void *id, int nranks, int rank);
int psh_comm_broadcast(void *buf_dev, size_t nbytes, int root);
int psh_comm_allgather(const void *send_dev, void *recv_dev, size_t nbytes_per_rank);
int psh_comm_allreduce_f32(float *buf_dev, size_t count, int op);
int psh_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* PYSTEPS_HIP_H */
