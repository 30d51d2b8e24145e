// Natural logarithm of a positive normal double, evaluated in double-double arithmetic and rounded
// once at the end (relative error of the unrounded value < 2^-80: the result is the correctly rounded
// logarithm except with probability ~1e-8 per argument).
//
// Why it exists: NumPy's legacy RandomState.randn (numpy/random/src/legacy/legacy-distributions.c,
// legacy_gauss; third party, numpy 2.2 installed) computes f = sqrt(-2 log(r2) / r2) with the C
// library's log().  glibc's log is faithful (< 0.52 ulp) but not correctly rounded (0.09 % of the
// arguments round the other way, tests/test_rng_cpu.py), and its tables are not ours to copy - so the
// device restatement of that stream (csrc/rng.hip) uses the one logarithm every platform can agree
// on, the correctly rounded one.  sqrt and the division are correctly rounded on both sides.
//
// Plain C++ (fma from <cmath>): the same source is compiled by hipcc for the kernels and by g++ for
// the host-side test of the rounding (tests/test_rng_cpu.py against decimal at 60 digits).
//
//   x = 2^e z, z in [0.75, 1.5);  i = round((z - 0.75) 256), c_i = 0.75 + i/256 (c_64 = 1);
//   z invc_i - 1 = r exactly as a double-double;  log x = e ln2 + logc_i + log1p(r),
//   log1p(r) = r + r^2 q(r), q by Horner - the three outer steps in double-double, the rest in double.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "cr_log_table.h"

#if defined(__HIPCC__)
#define PSH_CRLOG_FN __host__ __device__ __forceinline__
#else
#define PSH_CRLOG_FN inline
#endif
// The error-free transformations below are exact only if every product and sum is rounded on its
// own: a compiler that fuses `p = a * b` of two_prod with a later `p - 1.0` (hipcc's default,
// -ffp-contract=fast, does so across inlined calls) silently breaks them - 200 ulp in the first
// device run.  Fused operations are written as fma() where they are meant.
#if defined(__clang__)
#define PSH_CRLOG_EXACT _Pragma("clang fp contract(off)")
#else
#define PSH_CRLOG_EXACT
#endif

namespace psh {
namespace crlog {

struct dd {
  double hi, lo;
};

PSH_CRLOG_FN dd two_sum(double a, double b) {
  PSH_CRLOG_EXACT
  const double s = a + b;
  const double bb = s - a;
  return dd{s, (a - (s - bb)) + (b - bb)};
}
PSH_CRLOG_FN dd fast_two_sum(double a, double b) {  // |a| >= |b| or a == 0
  PSH_CRLOG_EXACT
  const double s = a + b;
  return dd{s, b - (s - a)};
}
PSH_CRLOG_FN dd two_prod(double a, double b) {
  PSH_CRLOG_EXACT
  const double p = a * b;
  return dd{p, fma(a, b, -p)};
}
PSH_CRLOG_FN dd add_dd(dd a, dd b) {
  PSH_CRLOG_EXACT
  dd s = two_sum(a.hi, b.hi);
  s.lo += a.lo + b.lo;
  return fast_two_sum(s.hi, s.lo);
}
PSH_CRLOG_FN dd mul_dd_d(dd a, double b) {
  PSH_CRLOG_EXACT
  dd p = two_prod(a.hi, b);
  p.lo = fma(a.lo, b, p.lo);
  return fast_two_sum(p.hi, p.lo);
}
PSH_CRLOG_FN dd mul_dd_dd(dd a, dd b) {
  PSH_CRLOG_EXACT
  dd p = two_prod(a.hi, b.hi);
  p.lo += a.hi * b.lo + a.lo * b.hi;
  return fast_two_sum(p.hi, p.lo);
}

// table: PSH_CRLOG_N rows {invc, logc_hi, logc_lo} (cr_log_table.h); the caller says where it lives
// (__constant__ memory on the device, a static array on the host)
PSH_CRLOG_FN double log_cr(double x, const double (*table)[3]) {
  PSH_CRLOG_EXACT
  uint64_t ix;
  memcpy(&ix, &x, 8);
  int e = static_cast<int>(ix >> 52) - 1023;
  ix = (ix & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
  double z;
  memcpy(&z, &ix, 8);  // [1, 2)
  if (z >= 1.5) {
    z *= 0.5;
    e += 1;
  }
  const int i = static_cast<int>((z - 0.75) * 256.0 + 0.5);
  const double invc = table[i][0];
  const dd p = two_prod(z, invc);
  const dd r = fast_two_sum(p.hi - 1.0, p.lo);  // z invc - 1, exact
  const double s = r.hi;

  // q(s) = -1/2 + s (1/3 + s (-1/4 + s v)),  v = 1/5 - s/6 + s^2/7 - ... + s^6/11
  double v = 1.0 / 11.0;
  v = fma(v, s, -1.0 / 10.0);
  v = fma(v, s, 1.0 / 9.0);
  v = fma(v, s, -1.0 / 8.0);
  v = fma(v, s, 1.0 / 7.0);
  v = fma(v, s, -1.0 / 6.0);
  dd t = two_prod(v, s);                                           // s (v - 1/5)
  t = add_dd(t, dd{PSH_CRLOG_FIFTH_HI, PSH_CRLOG_FIFTH_LO});       // v
  t = mul_dd_d(t, s);
  t = add_dd(t, dd{-0.25, 0.0});                                   // -1/4 + s v
  t = mul_dd_d(t, s);
  t = add_dd(t, dd{PSH_CRLOG_THIRD_HI, PSH_CRLOG_THIRD_LO});       // 1/3 + ...
  t = mul_dd_d(t, s);
  t = add_dd(t, dd{-0.5, 0.0});                                    // q(s)
  dd l = mul_dd_dd(two_prod(s, s), t);                             // s^2 q(s)
  l = add_dd(dd{s, 0.0}, l);                                       // log1p(s)
  // the low word of r: log1p(s + r.lo) = log1p(s) + r.lo / (1 + s) (+ O(r.lo^2))
  l.lo += r.lo * fma(fma(s, s, -s), 1.0, 1.0);
  l = fast_two_sum(l.hi, l.lo);

  dd acc = dd{table[i][1], table[i][2]};
  if (e != 0) {
    const double ed = static_cast<double>(e);
    dd k = dd{ed * PSH_CRLOG_LN2_HI, 0.0};  // exact: 32 significant bits times |e| < 2^11
    dd mid = two_prod(ed, PSH_CRLOG_LN2_MID);
    mid.lo = fma(ed, PSH_CRLOG_LN2_LO, mid.lo);
    k = add_dd(k, mid);
    acc = add_dd(k, acc);
  }
  acc = add_dd(acc, l);
  return acc.hi;  // add_dd renormalises: |lo| <= ulp(hi)/2, so hi is hi + lo rounded to nearest
}

}  // namespace crlog
}  // namespace psh
