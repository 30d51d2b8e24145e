/* CPU oracle (plain C, float64 arithmetic) for the semi-Lagrangian extrapolator.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/liboracle.so by
 * oracle/build.py and loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  The product (pysteps_amd) never links or
 * loads it.
 *
 * Restates pysteps/extrapolation/semilagrangian.py:200-255 (trajectory loop)
 * with the order-0/1 scipy.ndimage.map_coordinates semantics of SURVEY.md
 * section 8a row a3 (call sites semilagrangian.py:185-190, 225-232).  The trajectory
 * of a pixel depends only on gathers from the constant velocity field, so the
 * port walks every pixel through all lead steps independently (no full-grid
 * temporaries) and parallelises over rows with OpenMP.
 *
 * float32 inputs: map_coordinates allocates its output in the dtype of the
 * sampled array, so velocity samples are rounded to float32 before scaling
 * and advected values are rounded to float32 on store, like the reference.
 *
 * Pinned by tests/test_oracle_semilag.py against oracle/semilag.py, the
 * reference's two known-answer tests and the golden fixtures in tests/golden/.
 */
#include <math.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  long lo, hi;
  double frac;
} tap_t;

static inline tap_t taps(double c, long len) {
  tap_t t;
  double b = floor(c);
  t.frac = c - b;
  t.lo = (long)b;
  t.hi = t.lo + 1;
  if (t.hi > len - 1) t.hi = len >= 2 ? len - 2 : 0; /* mirrored, weight 0 */
  return t;
}

static inline double bilinear(const float *f, long n, tap_t r, tap_t c) {
  const double a = f[r.lo * n + c.lo], b = f[r.lo * n + c.hi];
  const double d = f[r.hi * n + c.lo], e = f[r.hi * n + c.hi];
  return (1.0 - r.frac) * (1.0 - c.frac) * a + (1.0 - r.frac) * c.frac * b +
         r.frac * (1.0 - c.frac) * d + r.frac * c.frac * e;
}

static inline double clampd(double v, double lo, double hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

/* both velocity components at (x+dx, y+dy), mode="nearest", rounded to f32 */
static inline void motion_at(const float *vel, long m, long n, long x, long y,
                             double dx, double dy, double scale, double sub,
                             double *ix, double *iy) {
  /* a non-finite coordinate (the trajectory met a non-finite velocity): SciPy interpolates
   * across it to NaN in mode "nearest" (sl_velnan* goldens of the reference) */
  if (!isfinite(dx) || !isfinite(dy)) {
    *ix = *iy = NAN;
    return;
  }
  const tap_t r = taps(clampd((double)y + dy, 0.0, (double)(m - 1)), m);
  const tap_t c = taps(clampd((double)x + dx, 0.0, (double)(n - 1)), n);
  const float u = (float)bilinear(vel, n, r, c);
  const float v = (float)bilinear(vel + (size_t)m * n, n, r, c);
  *ix = (double)u / sub * scale;
  *iy = (double)v / sub * scale;
}

/* steps[t] = (lead-time increment) / vel_timestep.  Returns 0. */
int oracle_semilag_f32(const float *precip, const float *vel, int m_, int n_,
                       const double *steps, int T, int n_iter, int order,
                       double outval, const double *disp_prev, float *out,
                       double *disp_out, int nthreads) {
  const long m = m_, n = n_;
  const double sub = n_iter > 1 ? (double)n_iter : 1.0;
  const size_t plane = (size_t)m * n;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
  for (long y = 0; y < m; ++y) {
    for (long x = 0; x < n; ++x) {
      const size_t p = (size_t)y * n + x;
      double dx, dy, ix, iy;
      if (disp_prev) {
        dx = disp_prev[p];
        dy = disp_prev[plane + p];
        motion_at(vel, m, n, x, y, dx, dy, steps[0], sub, &ix, &iy);
      } else {
        dx = dy = 0.0;
        ix = (double)vel[p] * steps[0]; /* not divided by n_iter: ref :202 */
        iy = (double)vel[plane + p] * steps[0];
      }
      for (int t = 0; t < T; ++t) {
        const double s = steps[t];
        if (n_iter > 0) {
          for (int k = 0; k < n_iter; ++k) {
            motion_at(vel, m, n, x, y, dx - ix / 2.0, dy - iy / 2.0, s, sub, &ix, &iy);
            dx -= ix;
            dy -= iy;
            motion_at(vel, m, n, x, y, dx, dy, s, sub, &ix, &iy);
          }
        } else {
          if (t > 0 || disp_prev) motion_at(vel, m, n, x, y, dx, dy, s, sub, &ix, &iy);
          dx -= ix;
          dy -= iy;
        }
        if (precip) {
          const double cy = (double)y + dy, cx = (double)x + dx;
          double val;
          /* non-finite coordinates count as outside (cval), like SciPy's mode "constant" */
          if (!isfinite(cy) || !isfinite(cx) || cy < 0.0 || cy > (double)(m - 1) || cx < 0.0 ||
              cx > (double)(n - 1)) {
            val = outval;
          } else if (order == 0) {
            long r = (long)floor(cy + 0.5), c = (long)floor(cx + 0.5);
            if (r > m - 1) r = m - 1;
            if (c > n - 1) c = n - 1;
            val = precip[r * n + c];
          } else {
            val = bilinear(precip, n, taps(cy, m), taps(cx, n));
          }
          out[(size_t)t * plane + p] = (float)val;
        }
      }
      if (disp_out) {
        disp_out[p] = dx;
        disp_out[plane + p] = dy;
      }
    }
  }
  return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
