// Two-dimensional FFTs in float64 for power-of-two grids, hand-written for gfx950.
//
// The FFT method object of pysteps (pysteps/utils/fft.py:20-37 get_numpy: fft2, ifft2, rfft2,
// irfft2 of numpy.fft, all in double precision) is what the STEPS member loop spends its time in
// once the advection is off the CPU: noise generation (pysteps/noise/fftgenerators.py:330-400),
// cascade decomposition (pysteps/cascade/decomposition.py:77-262) and the spectral recomposition
// run ~10 transforms of the full grid per member and lead time.  numpy's
// transforms live in pocketfft [third party, numpy 2.2]; what has to be reproduced is the
// mathematical transform with numpy's conventions (no scaling forward, 1/(m n) backward; rfft2
// keeps the n/2+1 non-negative frequencies of the last axis; irfft2 ignores the imaginary parts
// of the zero and Nyquist bins of the last axis), to round-off: parity bar rel-L2 <= 1e-12.
//
// Layout of a transform of an (m, n) grid, m and n powers of two:
//  * rows: one workgroup per PAIR of real rows - row a + i * row b is one complex FFT of length
//    n in LDS (padded against bank conflicts), the two spectra are separated on the way out
//    (A[k] = (Z[k] + conj Z[n-k]) / 2, B[k] = (Z[k] - conj Z[n-k]) / 2i).  Complex rows: one
//    workgroup per row.
//  * columns: one workgroup per 2 (m <= 4096) or 1 columns of the (m, n/2+1) / (m, n) complex
//    array, XCD-contiguous so that workgroups sharing cache lines share an L2.
//  * the FFT itself: decimation in time, bit-reversed on the way into LDS, two radix-2 layers per
//    pass (4 points per thread and pass: half the LDS traffic and barriers of plain radix 2),
//    twiddles exp(-2 pi i k / N) from a table computed once per length on the host in long double.
// FP64 vector rate on MI355X equals FP32 (78 TFLOP/s); a 4096^2 rfft2 is ~0.5 GFLOP: the transform
// is bound by HBM / LDS traffic, not arithmetic (docs/history.md 3.6).
//
// Sides that are NOT powers of two (radar composites: 640 x 710, 1226 x 760 ...; the reference's FFT
// object takes any shape, pysteps/utils/fft.py:20-37) go through Bluestein's chirp-z identity inside
// the same kernels: with b_j = exp(i pi j^2 / N)
//     X_k = conj(b_k) sum_j (x_j conj(b_j)) b_(k-j)
// is a circular convolution of length M = 2^ceil(log2(2N-1)), i.e. two of the power-of-two LDS
// transforms above with a pointwise product in between (the spectrum of the chirp is a table per
// length, computed once on the device).  One side up to 4096 (M <= 8192 = the LDS transform limit).
#include <cmath>
#include <cstdlib>
#include <map>
#include <vector>

#include "common.h"

namespace psh {
namespace {

constexpr int kFftThreads = 1024;  // launch bound; the launchers pick the count by transform length:
// one transform wants ~length / 8 threads (measured: 2048 -> 256, 4096 -> 512, 8192 -> 1024 threads;
// fewer leave the passes latency-bound, more only add barrier cost)
inline int fft_threads(int points) {
  int t = points / 8;
  t = t < 64 ? 64 : t > kFftThreads ? kFftThreads : t;
  return (t + 63) & ~63;
}
constexpr int kFftMaxLog = 13;  // 8192 points: 128 KiB of LDS

// LDS index of element i: the low four bits (the 16-byte unit inside a 256-byte LDS row) are mixed
// with higher index bits, so that the strided accesses of the passes (4^s apart), of the bit-reversed
// store and of the n-k mirror spread over the banks.  Found by a search over XOR swizzles against all
// access patterns of lengths 2^10..2^13 (tools/fft_swizzle.py): 1.33 16-byte units per bank and
// 16-lane group on average, against 2.2 for padding i + (i >> 6); a bijection on every aligned
// block of 2^12 elements, no padding.
__device__ __forceinline__ int lpad(int i) { return i ^ (((i >> 4) ^ (i >> 5) ^ (i >> 9)) & 15); }
__host__ __device__ constexpr int lds_elems(int n) { return n < 16 ? 16 : n; }

__device__ __forceinline__ double2 cmul(double2 a, double2 w) {
  return make_double2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

// one butterfly of a pass (two radix-2 layers): wb = W_4h^j (already conjugated for INV); the results
// replace x0..x3 at the positions they were read from (e0, e0 + h, e0 + 2h, e0 + 3h)
template <bool INV>
__device__ __forceinline__ void radix4(double2 &x0, double2 &x1, double2 &x2, double2 &x3, double2 wb) {
  const double2 wa = cmul(wb, wb);                                                  // W_2h^j
  const double2 wc = INV ? make_double2(-wb.y, wb.x) : make_double2(wb.y, -wb.x);  // W_4h^(j+h)
  const double2 t1 = cmul(x1, wa), t3 = cmul(x3, wa);
  const double2 a0 = make_double2(x0.x + t1.x, x0.y + t1.y), a1 = make_double2(x0.x - t1.x, x0.y - t1.y);
  const double2 a2 = make_double2(x2.x + t3.x, x2.y + t3.y), a3 = make_double2(x2.x - t3.x, x2.y - t3.y);
  const double2 u2 = cmul(a2, wb), u3 = cmul(a3, wc);
  x0 = make_double2(a0.x + u2.x, a0.y + u2.y);
  x2 = make_double2(a0.x - u2.x, a0.y - u2.y);
  x1 = make_double2(a1.x + u3.x, a1.y + u3.y);
  x3 = make_double2(a1.x - u3.x, a1.y - u3.y);
}

// TWO consecutive passes (four radix-2 layers, half sizes h and 4h) on 16 elements e0 + a h + b 4h held in
// registers in between: the first pass couples a (same twiddle W_4h^j for every b), the second couples b
// (twiddle W_16h^(j + a h)).  The same operations on the same operands as two separate passes, with one
// LDS round trip, one barrier and one set of index arithmetic instead of two - the passes are bound by
// VALU issue (35 instructions per point and pass, 10 of them arithmetic: profiles/r03/n_fft_pmc.csv).
template <bool INV>
__device__ __forceinline__ void fft_pass16(double2 *z, int pitch, int count, int logn, int s,
                                           const double2 *__restrict__ tw, int tws) {
  const int N = 1 << logn, h = 1 << s;
  const int st_a = N >> (s + 2), st_b = N >> (s + 4);  // W_4h^j = W_N^(j st_a), W_16h^j' = W_N^(j' st_b)
  const int total = (N >> 4) * count;
  for (int t = threadIdx.x; t < total; t += static_cast<int>(blockDim.x)) {
    const int c = t >> (logn - 4), q = t & ((N >> 4) - 1);
    const int j = q & (h - 1);
    const int e0 = ((q >> s) << (s + 4)) + j;
    const int base = c * pitch;
    // lpad is linear over GF(2) and the bits of a h + b 4h are zero in e0 (no carries): the swizzled slot of
    // e0 + a h + b 4h is lpad(e0) ^ lpad(a h + b 4h), the second factor the same for the whole wave
    const int slot0 = lpad(e0);
    int at[4][4];
    double2 x[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        at[a][b] = base + (slot0 ^ lpad(a * h + b * 4 * h));
        x[a][b] = z[at[a][b]];
      }
    }
    double2 wa = tw[j * st_a * tws];
    double2 wb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) wb[a] = tw[(j + a * h) * st_b * tws];
    if (INV) {
      wa.y = -wa.y;
#pragma unroll
      for (int a = 0; a < 4; ++a) wb[a].y = -wb[a].y;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) radix4<INV>(x[0][b], x[1][b], x[2][b], x[3][b], wa);
#pragma unroll
    for (int a = 0; a < 4; ++a) radix4<INV>(x[a][0], x[a][1], x[a][2], x[a][3], wb[a]);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
      for (int a = 0; a < 4; ++a) z[at[a][b]] = x[a][b];
    }
  }
  __syncthreads();
}

// in-place DIT FFT of `count` independent sequences z[c * pitch + lpad(i)], i < N = 1 << logn, whose
// elements were stored bit-reversed; INV conjugates the twiddles (unscaled inverse).
// A thread takes four butterflies per round: their sixteen LDS reads and four twiddle loads are
// issued before any arithmetic and their sixteen writes after it (the butterflies of a pass touch
// disjoint elements) - written as one loop body, the compiler has to assume that a butterfly's
// writes alias the next one's reads and serialises read -> compute -> write with the LDS / L2
// latency exposed every time (the first version: 12 us per 4096-point transform instead of ~3).
// Only W_4h^j is fetched per butterfly: W_2h^j is its square and W_4h^(j+h) = -i W_4h^j.
// `tw` is the table of a length N * tws (entry k * tws = exp(-2 pi i k / N)): sub-transforms of a
// longer transform read the long table with a stride.
// (Compile-time lengths - every shift, mask and twiddle stride an immediate, the pass loop unrolled - were
// measured 2 % faster and cost four copies of every kernel: not kept.)
template <bool INV>
__device__ __forceinline__ void fft_lds(double2 *z, int pitch, int count, int logn,
                                        const double2 *__restrict__ tw, int tws = 1) {
  const int N = 1 << logn;
  constexpr int kU = 4;  // butterflies in flight per thread
  int s = 0;
  if (logn & 1) {  // a single radix-2 layer first (twiddle 1), then pairs of layers
    const int total = (N >> 1) * count;
    for (int b0 = threadIdx.x; b0 < total; b0 += kU * static_cast<int>(blockDim.x)) {
      int at[kU];
      double2 x0[kU], x1[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int b = b0 + u * static_cast<int>(blockDim.x);
        const int c = b >> (logn - 1), q = b & ((N >> 1) - 1);
        at[u] = b < total ? c * pitch + lpad(q << 1) : -1;
        if (at[u] >= 0) {
          x0[u] = z[at[u]];
          x1[u] = z[at[u] ^ 1];  // lpad(2q + 1) = lpad(2q) ^ 1: the swizzle only reads bits >= 4
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (at[u] < 0) continue;
        z[at[u]] = make_double2(x0[u].x + x1[u].x, x0[u].y + x1[u].y);
        z[at[u] ^ 1] = make_double2(x0[u].x - x1[u].x, x0[u].y - x1[u].y);
      }
    }
    __syncthreads();
    s = 1;
  }
  for (; s + 4 <= logn; s += 4) fft_pass16<INV>(z, pitch, count, logn, s, tw, tws);
  for (; s < logn; s += 2) {  // (logn - s is even here): the last pair of layers where the count is not a multiple of four
    const int h = 1 << s;          // half size of the first layer
    const int st2 = N >> (s + 2);  // W_{4h}^j = W_N^{j st2}
    const int total = (N >> 2) * count;
    for (int b0 = threadIdx.x; b0 < total; b0 += kU * static_cast<int>(blockDim.x)) {
      int i0[kU], i1[kU], i2[kU], i3[kU];
      double2 x0[kU], x1[kU], x2[kU], x3[kU], w2[kU];
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int b = b0 + u * static_cast<int>(blockDim.x);
        const bool live = b < total;
        const int c = b >> (logn - 2), q = b & ((N >> 2) - 1);
        const int j = q & (h - 1);
        const int e0 = ((q >> s) << (s + 2)) + j;
        const int base = c * pitch;
        i0[u] = live ? base + lpad(e0) : -1;
        i1[u] = base + lpad(e0 + h);
        i2[u] = base + lpad(e0 + 2 * h);
        i3[u] = base + lpad(e0 + 3 * h);
        if (live) {
          w2[u] = tw[j * st2 * tws];
          x0[u] = z[i0[u]];
          x1[u] = z[i1[u]];
          x2[u] = z[i2[u]];
          x3[u] = z[i3[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (i0[u] < 0) continue;
        double2 wb = w2[u];
        if (INV) wb.y = -wb.y;
        const double2 wa = cmul(wb, wb);                                                  // W_2h^j
        const double2 wc = INV ? make_double2(-wb.y, wb.x) : make_double2(wb.y, -wb.x);  // W_4h^(j+h)
        const double2 t1 = cmul(x1[u], wa), t3 = cmul(x3[u], wa);
        const double2 a0 = make_double2(x0[u].x + t1.x, x0[u].y + t1.y), a1 = make_double2(x0[u].x - t1.x, x0[u].y - t1.y);
        const double2 a2 = make_double2(x2[u].x + t3.x, x2[u].y + t3.y), a3 = make_double2(x2[u].x - t3.x, x2[u].y - t3.y);
        const double2 u2 = cmul(a2, wb), u3 = cmul(a3, wc);
        z[i0[u]] = make_double2(a0.x + u2.x, a0.y + u2.y);
        z[i2[u]] = make_double2(a0.x - u2.x, a0.y - u2.y);
        z[i1[u]] = make_double2(a1.x + u3.x, a1.y + u3.y);
        z[i3[u]] = make_double2(a1.x - u3.x, a1.y - u3.y);
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int i, int logn) { return static_cast<int>(__brev(static_cast<unsigned>(i)) >> (32 - logn)); }

// ---- any length: Bluestein on top of fft_lds -------------------------------------------------------
// Dft describes the 1-D transform of one axis: a plain power-of-two transform (chirp == nullptr) or a
// chirp-z transform of length n inside LDS sequences of M = 1 << logm points.
struct Dft {
  int n;                  // transform length
  int logm;               // LDS transform: log2 of n (plain) or of M (Bluestein)
  const double2 *tw;      // twiddles of the LDS transform length
  const double2 *chirp;   // b_j = exp(+i pi j^2 / n), j < n; nullptr: plain transform
  const double2 *filter;  // FFT_M of the symmetric extension of b (b_j at j and M - j)
};

__device__ __forceinline__ double2 cmulc(double2 a, double2 w) {  // a * conj(w)
  return make_double2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
}

// LDS slot of input sample i (callers store x_i there, pre-multiplied by pre_chirp; Bluestein
// sequences must hold zeros at the slots of i = n .. M-1)
__device__ __forceinline__ int dft_in_slot(const Dft &d, int i) { return lpad(bitrev(i, d.logm)); }
template <bool INV>
__device__ __forceinline__ double2 dft_pre(const Dft &d, int i, double2 v) {
  if (!d.chirp) return v;
  return INV ? cmul(v, d.chirp[i]) : cmulc(v, d.chirp[i]);
}
// output k of sequence c after dft_lds (unscaled: an inverse transform still wants 1 / n)
template <bool INV>
__device__ __forceinline__ double2 dft_out(const Dft &d, const double2 *z, int base, int k) {
  const double2 v = z[base + lpad(k)];
  if (!d.chirp) return v;
  const double inv_m = 1.0 / static_cast<double>(1 << d.logm);
  const double2 w = INV ? cmul(v, d.chirp[k]) : cmulc(v, d.chirp[k]);
  return make_double2(w.x * inv_m, w.y * inv_m);
}

template <bool INV>
__device__ __forceinline__ void dft_lds(double2 *z, int pitch, int count, const Dft &d) {
  if (!d.chirp) {
    fft_lds<INV>(z, pitch, count, d.logm, d.tw);
    return;
  }
  fft_lds<false>(z, pitch, count, d.logm, d.tw);
  // spectrum x spectrum of the chirp, put back bit-reversed for the second transform: element k
  // and element rev(k) trade places (one thread per pair)
  const int M = 1 << d.logm;
  for (int idx = threadIdx.x; idx < count * M; idx += blockDim.x) {
    const int c = idx >> d.logm, k = idx & (M - 1);
    const int r = bitrev(k, d.logm);
    if (k > r) continue;
    const int base = c * pitch;
    double2 fk = d.filter[k], fr = d.filter[r];
    if (INV) {  // the chirp of the inverse transform is the conjugate one; its extension is symmetric
      fk.y = -fk.y;
      fr.y = -fr.y;
    }
    const double2 vk = cmul(z[base + lpad(k)], fk);
    if (k == r) {
      z[base + lpad(k)] = vk;
    } else {
      const double2 vr = cmul(z[base + lpad(r)], fr);
      z[base + lpad(k)] = vr;
      z[base + lpad(r)] = vk;
    }
  }
  __syncthreads();
  fft_lds<true>(z, pitch, count, d.logm, d.tw);
}

// ---- real rows -> half spectra: two rows per workgroup -----------------------------------------
__global__ __launch_bounds__(kFftThreads) void fft_rows_r2c(const double *__restrict__ x, int m, Dft d,
                                                            double2 *__restrict__ out) {
  extern __shared__ double2 z[];
  const int n = d.n, len = 1 << d.logm;
  const int ra = 2 * blockIdx.x, rb = min(ra + 1, m - 1);
  const int nc = n / 2 + 1;
  const double *xa = x + static_cast<size_t>(ra) * n, *xb = x + static_cast<size_t>(rb) * n;
  for (int i = threadIdx.x; i < len; i += blockDim.x)
    z[dft_in_slot(d, i)] = i < n ? dft_pre<false>(d, i, make_double2(xa[i], xb[i])) : make_double2(0.0, 0.0);
  __syncthreads();
  dft_lds<false>(z, 0, 1, d);
  double2 *oa = out + static_cast<size_t>(ra) * nc, *ob = out + static_cast<size_t>(rb) * nc;
  for (int k = threadIdx.x; k < nc; k += blockDim.x) {
    const double2 zk = dft_out<false>(d, z, 0, k), zn = dft_out<false>(d, z, 0, k == 0 ? 0 : n - k);
    oa[k] = make_double2(0.5 * (zk.x + zn.x), 0.5 * (zk.y - zn.y));
    if (ra + 1 < m) ob[k] = make_double2(0.5 * (zk.y + zn.y), -0.5 * (zk.x - zn.x));
  }
}

// ---- half spectra -> real rows (numpy irfft: the imaginary parts of bins 0 and n/2 are ignored) --
// min_out (may be nullptr): np.min of the whole output as an order-preserving key (NaN -> 0, the smallest key), folded
// in with one atomic per wave - the STEPS member update needs the field's minimum right after the transform
// (steps_loop.hip field_min_key: a sweep of its own over 8 B per pixel otherwise)
__device__ __forceinline__ unsigned long long fft_min_key(double v) {
  if (v != v) return 0ull;
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(v));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ __launch_bounds__(kFftThreads) void fft_rows_c2r(const double2 *__restrict__ in, int m, Dft d, double scale,
                                                            double *__restrict__ out, unsigned long long *__restrict__ min_out) {
  extern __shared__ double2 z[];
  const int n = d.n, len = 1 << d.logm;
  const int ra = 2 * blockIdx.x, rb = min(ra + 1, m - 1);
  const int nc = n / 2 + 1;
  const int nyq = (n & 1) ? -1 : n / 2;  // odd lengths have no Nyquist bin
  const double2 *ia = in + static_cast<size_t>(ra) * nc, *ib = in + static_cast<size_t>(rb) * nc;
  if (d.chirp) {  // the slots beyond n stay zero
    for (int i = n + threadIdx.x; i < len; i += blockDim.x) z[dft_in_slot(d, i)] = make_double2(0.0, 0.0);
  }
  for (int k = threadIdx.x; k < nc; k += blockDim.x) {
    double2 a = ia[k], b = ib[k];
    if (k == 0 || k == nyq) {
      a.y = 0.0;
      b.y = 0.0;
    }
    // Z[k] = A[k] + i B[k];  Z[n-k] = conj(A[k]) + i conj(B[k])
    z[dft_in_slot(d, k)] = dft_pre<true>(d, k, make_double2(a.x - b.y, a.y + b.x));
    if (k != 0 && k != nyq) z[dft_in_slot(d, n - k)] = dft_pre<true>(d, n - k, make_double2(a.x + b.y, b.x - a.y));
  }
  __syncthreads();
  dft_lds<true>(z, 0, 1, d);
  double *oa = out + static_cast<size_t>(ra) * n, *ob = out + static_cast<size_t>(rb) * n;
  unsigned long long key = ~0ull;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const double2 v = dft_out<true>(d, z, 0, i);
    const double a = v.x * scale, b = v.y * scale;
    oa[i] = a;
    if (ra + 1 < m) ob[i] = b;
    if (min_out) {
      const unsigned long long ka = fft_min_key(a), kb = ra + 1 < m ? fft_min_key(b) : ~0ull;
      key = ka < key ? ka : key;
      key = kb < key ? kb : key;
    }
  }
  if (min_out) {  // (uniform)
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      const unsigned long long o = __shfl_xor(key, s);
      key = o < key ? o : key;
    }
    if ((threadIdx.x & 63) == 0 && key < __hip_atomic_load(min_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(min_out, key);
  }
}

// ---- complex rows: one row per workgroup ---------------------------------------------------------
template <bool INV>
__global__ __launch_bounds__(kFftThreads) void fft_rows_c2c(const double2 *__restrict__ in, Dft d,
                                                            double2 *__restrict__ out) {
  extern __shared__ double2 z[];
  const int n = d.n, len = 1 << d.logm;
  const double2 *src = in + static_cast<size_t>(blockIdx.x) * n;
  for (int i = threadIdx.x; i < len; i += blockDim.x)
    z[dft_in_slot(d, i)] = i < n ? dft_pre<INV>(d, i, src[i]) : make_double2(0.0, 0.0);
  __syncthreads();
  dft_lds<INV>(z, 0, 1, d);
  double2 *dst = out + static_cast<size_t>(blockIdx.x) * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = dft_out<INV>(d, z, 0, i);
}

// ---- columns of an (m, nc) complex array: `cols` adjacent columns per workgroup ---------------------
template <bool INV>
__global__ __launch_bounds__(kFftThreads) void fft_cols_c2c(const double2 *__restrict__ in, Dft d, int nc,
                                                            int cols, double scale,
                                                            double2 *__restrict__ out, int groups,
                                                            int groups_per_xcd, const double *__restrict__ weights) {
  extern __shared__ double2 z[];
  // XCD-contiguous column groups: neighbours share 128-byte lines of every row
  const int b = blockIdx.x;
  const int g = (b % kNumXcd) * groups_per_xcd + b / kNumXcd;
  if (g >= groups) return;
  const int m = d.n, len = 1 << d.logm;
  const int c0 = g * cols;
  const int live = min(cols, nc - c0);
  const int pitch = lds_elems(len);
  // (Starting every group's walk down its columns at its own row - against all workgroups asking for
  // the same rows, 32 KiB + 16 B apart, at the same time - was measured 7 % SLOWER at 4096^2: groups
  // that share 128-byte lines then no longer ask for them together, and the lines leave the L2 in between.)
  for (int idx = threadIdx.x; idx < len * cols; idx += blockDim.x) {
    const int r = idx / cols, c = idx - r * cols;
    double2 v = make_double2(0.0, 0.0);
    if (c < live && r < m) {
      v = in[static_cast<size_t>(r) * nc + c0 + c];
      if (weights) {  // spectrum x real filter (band-pass weights, noise filter) applied on the way in
        const double w = weights[static_cast<size_t>(r) * nc + c0 + c];
        v.x *= w;
        v.y *= w;
      }
      v = dft_pre<INV>(d, r, v);
    }
    z[c * pitch + dft_in_slot(d, r)] = v;
  }
  __syncthreads();
  dft_lds<INV>(z, pitch, cols, d);
  for (int idx = threadIdx.x; idx < m * cols; idx += blockDim.x) {
    const int r = idx / cols, c = idx - r * cols;
    if (c < live) {
      const double2 v = dft_out<INV>(d, z, c * pitch, r);
      out[static_cast<size_t>(r) * nc + c0 + c] = make_double2(v.x * scale, v.y * scale);
    }
  }
}

// ---- long columns in two sweeps (four-step FFT) ------------------------------------------------------
// fft_cols_c2c holds whole columns in LDS: 2 x 4096 x 16 B per workgroup means 32-byte pieces of 4096
// different rows - one L1 miss per row and column pair, and the pass is bound by misses in flight x
// latency (190 us for 134 MB in and out at 4096^2, whatever the occupancy:
// profiles/r03/i_fft_column_pass_probe.txt).  With N = N1 N2, n = n1 + N1 n2, k = N2 k1 + k2
//     X[N2 k1 + k2] = sum_n1 W_N1^(n1 k1) [ W_N^(n1 k2) sum_n2 x[n1 + N1 n2] W_N2^(n2 k2) ]
// a column transform is N1 transforms of length N2 over rows N1 apart (sweep A, which also applies
// the twiddles W_N^(n1 k2) and leaves Z[n1, k2] at row n1 + N1 k2) followed by N2 transforms of length
// N1 over N1 CONSECUTIVE rows (sweep B, output row N2 k1 + k2).  A workgroup then needs only 512
// elements of a column at a time and takes EIGHT columns instead: every row it touches is a
// 128-byte request.  Twice the traffic of the one-sweep pass, all of it in full lines.
// a workgroup holds 2^LOGT elements (16 B each): 2^LOGC columns x 2^(LOGT - LOGC) elements of each, one thread per 8 elements

__device__ __forceinline__ double2 root_of_unity(const double2 *__restrict__ tw, int p, int half, bool inv) {
  double2 w = tw[p >= half ? p - half : p];  // W_N^p, p < N: W^(p + N/2) = -W^p
  if (p >= half) w = make_double2(-w.x, -w.y);
  if (inv) w.y = -w.y;
  return w;
}

// SWEEP 0 (A): sequences (column c, n1), elements n2, rows n1 + N1 n2 -> Z at rows n1 + N1 k2
// SWEEP 1 (B): sequences (column c, k2), elements n1, rows n1 + N1 k2 -> X at rows N2 k1 + k2
template <bool INV, int SWEEP, int LOGC, int LOGT>
__global__ __launch_bounds__(1 << (LOGT - 3)) void fft_cols_step(const double2 *__restrict__ in, const double2 *__restrict__ tw,
                                                              int logn, int log1, int nc, double scale,
                                                              double2 *__restrict__ out, int items, int items_per_xcd,
                                                              int col_tiles, const double *__restrict__ weights) {
  extern __shared__ double2 z[];
  const int b = blockIdx.x;
  const int item = (b % kNumXcd) * items_per_xcd + b / kNumXcd;  // neighbouring column tiles share an L2
  if (item >= items) return;
  const int tile = item % col_tiles, group = item / col_tiles;
  const int log2_ = logn - log1;                      // N2 = 1 << log2_
  const int N1 = 1 << log1, N2 = 1 << log2_;
  const int logl = SWEEP == 0 ? log2_ : log1;         // length of this sweep's transforms
  const int L = 1 << logl;
  constexpr int kStepCols = 1 << LOGC, kStepElems = 1 << (LOGT - LOGC), kStepThreads = 1 << (LOGT - 3);
  const int per_col = kStepElems >> logl;             // sequences per column in this workgroup
  const int pitch = L + 2;  // sequences two 16-byte units apart in the banks (the tile is walked across sequences; even: fft_lds pairs slots by ^ 1)
  const int first = group * per_col;                  // first n1 (A) / k2 (B) of the group
  const int c0 = tile * kStepCols;
  const int live = min(kStepCols, nc - c0);
  // element e of the tile: column e % 8, then (sequence, element) so that consecutive threads read one row
  for (int idx = threadIdx.x; idx < kStepCols * kStepElems; idx += kStepThreads) {
    const int c = idx & (kStepCols - 1), q = idx >> LOGC;
    int seq, el, row;
    if (SWEEP == 0) {
      seq = q & (per_col - 1);  // n1 - first: rows first .. first + per_col - 1 are consecutive
      el = q / per_col;         // n2
      row = first + seq + (el << log1);
    } else {
      el = q & (L - 1);         // n1: consecutive rows
      seq = q >> logl;          // k2 - first
      row = el + ((first + seq) << log1);
    }
    double2 v = make_double2(0.0, 0.0);
    if (c < live) {
      const size_t at = static_cast<size_t>(row) * nc + c0 + c;
      v = in[at];
      if (SWEEP == 0 && weights) {
        const double w = weights[at];
        v.x *= w;
        v.y *= w;
      }
    }
    z[(c * per_col + seq) * pitch + lpad(bitrev(el, logl))] = v;
  }
  __syncthreads();
  fft_lds<INV>(z, pitch, kStepCols * per_col, logl, tw, SWEEP == 0 ? N1 : N2);
  for (int idx = threadIdx.x; idx < kStepCols * kStepElems; idx += kStepThreads) {
    const int c = idx & (kStepCols - 1), q = idx >> LOGC;
    int seq, k, row;
    if (SWEEP == 0) {
      seq = q & (per_col - 1);
      k = q / per_col;          // k2
      row = first + seq + (k << log1);
    } else {
      seq = q & (per_col - 1);  // k2 - first: rows N2 k1 + k2 of a group are consecutive
      k = q / per_col;          // k1
      row = (k << log2_) + first + seq;
    }
    if (c >= live) continue;
    double2 v = z[(c * per_col + seq) * pitch + lpad(k)];
    if (SWEEP == 0) {
      v = cmul(v, root_of_unity(tw, (first + seq) * k, 1 << (logn - 1), INV));
    } else {
      v = make_double2(v.x * scale, v.y * scale);
    }
    out[static_cast<size_t>(row) * nc + c0 + c] = v;
  }
}

int ilog2_exact(int v) {
  if (v < 2 || (v & (v - 1)) != 0) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// twiddle table of a length, device resident, computed once (lock held by the callers)
std::map<int, double2 *> &twiddle_cache() {
  static std::map<int, double2 *> cache;
  return cache;
}

int twiddles(int n, const double2 **tw_out) {
  std::map<int, double2 *> &cache = twiddle_cache();
  auto it = cache.find(n);
  if (it != cache.end()) {
    *tw_out = it->second;
    return PSH_OK;
  }
  const int count = n / 2 > 0 ? n / 2 : 1;
  std::vector<double2> host(static_cast<size_t>(count));
  const long double step = -2.0L * 3.14159265358979323846264338327950288L / static_cast<long double>(n);
  for (int k = 0; k < count; ++k) {
    const long double a = step * static_cast<long double>(k);
    host[k] = make_double2(static_cast<double>(cosl(a)), static_cast<double>(sinl(a)));
  }
  void *dev = nullptr;
  PSH_HIP(hipMalloc(&dev, host.size() * sizeof(double2)));
  const hipError_t e = hipMemcpy(dev, host.data(), host.size() * sizeof(double2), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(dev);
    return fail(PSH_EHIP, "fft: twiddle upload failed: %s", hipGetErrorString(e));
  }
  cache[n] = static_cast<double2 *>(dev);
  *tw_out = cache[n];
  return PSH_OK;
}

template <class K>
int allow_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    PSH_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(bytes)));
  return PSH_OK;
}

constexpr int kBluesteinMax = 1 << (kFftMaxLog - 1);  // 2 n - 1 <= 8192

// chirp b_j = exp(i pi j^2 / n) and the spectrum of its symmetric extension, per length
struct ChirpTables {
  double2 *chirp, *filter;
};
std::map<int, ChirpTables> &chirp_cache() {
  static std::map<int, ChirpTables> cache;
  return cache;
}

// the 1-D transform of one axis (lock held by the callers)
int make_dft(const char *who, int n, Dft *d) {
  d->n = n;
  d->chirp = d->filter = nullptr;
  const int lg = ilog2_exact(n);
  if (lg >= 1 && lg <= kFftMaxLog) {
    d->logm = lg;
    return twiddles(n, &d->tw);
  }
  if (n < 2 || n > kBluesteinMax)
    return fail(PSH_EUNSUPPORTED, "%s: side %d - powers of two up to %d, other lengths in 2..%d", who, n, 1 << kFftMaxLog,
                kBluesteinMax);
  int logm = 1;
  while ((1 << logm) < 2 * n - 1) ++logm;
  const int M = 1 << logm;
  d->logm = logm;
  if (int rc = twiddles(M, &d->tw)) return rc;
  std::map<int, ChirpTables> &cache = chirp_cache();
  auto it = cache.find(n);
  if (it == cache.end()) {
    // j^2 is reduced modulo 2 n exactly before the angle is formed: b_j has period 2 n in j^2
    std::vector<double2> b(static_cast<size_t>(n)), ext(static_cast<size_t>(M), make_double2(0.0, 0.0));
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int j = 0; j < n; ++j) {
      const long long q = (static_cast<long long>(j) * j) % (2LL * n);
      const long double a = pi * static_cast<long double>(q) / static_cast<long double>(n);
      b[j] = make_double2(static_cast<double>(cosl(a)), static_cast<double>(sinl(a)));
      ext[j] = b[j];
      if (j) ext[M - j] = b[j];
    }
    ChirpTables t{nullptr, nullptr};
    PSH_HIP(hipMalloc(reinterpret_cast<void **>(&t.chirp), b.size() * sizeof(double2)));
    PSH_HIP(hipMalloc(reinterpret_cast<void **>(&t.filter), ext.size() * sizeof(double2)));
    PSH_HIP(hipMemcpy(t.chirp, b.data(), b.size() * sizeof(double2), hipMemcpyHostToDevice));
    PSH_HIP(hipMemcpy(t.filter, ext.data(), ext.size() * sizeof(double2), hipMemcpyHostToDevice));
    // its spectrum: one plain row transform of length M, in place
    Dft plain{M, logm, d->tw, nullptr, nullptr};
    const size_t lds = static_cast<size_t>(lds_elems(M)) * sizeof(double2);
    if (int rc = allow_lds(fft_rows_c2c<false>, lds)) return rc;
    hipStream_t stream = ctx().stream;
    hipLaunchKernelGGL(fft_rows_c2c<false>, dim3(1), dim3(fft_threads(M)), lds, stream, t.filter, plain, t.filter);
    PSH_HIP(hipGetLastError());
    PSH_HIP(hipStreamSynchronize(stream));
    it = cache.emplace(n, t).first;
  }
  d->chirp = it->second.chirp;
  d->filter = it->second.filter;
  return PSH_OK;
}

int check_shape(const char *who, int m, int n, Dft *rows, Dft *cols) {
  if (int rc = make_dft(who, n, rows)) return rc;
  return make_dft(who, m, cols);
}

int launch_cols(bool inverse, const double2 *in, const Dft &d, int nc, double scale, double2 *out,
                hipStream_t stream, const double *weights = nullptr) {
  const int len = 1 << d.logm;
  static const int four_step = [] { const char *e = std::getenv("PYSTEPS_HIP_FFT_FOURSTEP"); return e ? std::atoi(e) : 1; }();
  // measured (profiles/r03/i_fft_column_pass_probe.txt): 11 % faster than one sweep at 8192 points, equal at
  // 4096, 10-25 % slower at 1024 / 2048 (PYSTEPS_HIP_FFT_FOURSTEP=2 takes it from 1024 points on)
  if (four_step && !d.chirp && d.logm >= (four_step == 2 ? 10 : 13)) {
    // two sweeps over column tiles (fft_cols_step) through a block of the same size
    void *tmp = nullptr;
    if (int rc = psh_malloc(&tmp, static_cast<size_t>(len) * nc * sizeof(double2))) return rc;
    const int log1 = d.logm / 2, log2_ = d.logm - log1;
    auto sweep = [&](auto kernel, int lc, int lt, int logl, int sequences, const double2 *src, double2 *dst, double sc,
                     const double *w) -> int {
      const int cols = 1 << lc, elems = 1 << (lt - lc);
      const int col_tiles = (nc + cols - 1) / cols;
      const int per_col = elems >> logl;
      if (per_col < 1 || sequences % per_col != 0) return fail(PSH_EINVAL, "fft: column tile does not fit the transform");
      const int items = col_tiles * (sequences / per_col);
      const int ipx = (items + kNumXcd - 1) / kNumXcd;
      const size_t lds = static_cast<size_t>(cols) * per_col * ((1 << logl) + 2) * sizeof(double2);
      if (int rc = allow_lds(kernel, lds)) return rc;
      hipLaunchKernelGGL(kernel, dim3(ipx * kNumXcd), dim3(1 << (lt - 3)), lds, stream, src, d.tw, d.logm, log1, nc, sc, dst,
                         items, ipx, col_tiles, w);
      PSH_HIP(hipGetLastError());
      return PSH_OK;
    };
    int rc = PSH_OK;
    double2 *mid = static_cast<double2 *>(tmp);
#define PSH_STEPS(LC, LT)                                                                                               \
  rc = inverse ? sweep(fft_cols_step<true, 0, LC, LT>, LC, LT, log2_, 1 << log1, in, mid, 1.0, weights)                 \
               : sweep(fft_cols_step<false, 0, LC, LT>, LC, LT, log2_, 1 << log1, in, mid, 1.0, weights);               \
  if (rc == PSH_OK)                                                                                                     \
    rc = inverse ? sweep(fft_cols_step<true, 1, LC, LT>, LC, LT, log1, 1 << log2_, mid, out, scale, nullptr)            \
                 : sweep(fft_cols_step<false, 1, LC, LT>, LC, LT, log1, 1 << log2_, mid, out, scale, nullptr)
    // 16 columns x 128 elements per workgroup of 256 threads: tiles of 2^10 .. 2^12 elements and 8 .. 32 columns were
    // measured within 4 % of each other (profiles/r03/i_fft_column_pass_probe.txt), this one the fastest at 8192 points
    PSH_STEPS(4, 11);
#undef PSH_STEPS
    (void)psh_free(tmp);  // stream-ordered
    return rc;
  }
  // development knobs (tools/fft_quick.py): columns per workgroup / threads per workgroup of this pass
  static const int forced_cols = [] { const char *e = std::getenv("PYSTEPS_HIP_FFT_COLS"); return e ? std::atoi(e) : 0; }();
  static const int forced_threads = [] { const char *e = std::getenv("PYSTEPS_HIP_FFT_COL_THREADS"); return e ? std::atoi(e) : 0; }();
  // two columns per workgroup share 32-byte sectors; at 1024 and 2048 points one column per workgroup
  // of 256 threads (more workgroups per CU) measured 16 % / 6 % faster, at 4096 the two are equal
  const bool narrow = len == 1024 || len == 2048;
  const int cols = forced_cols > 0 && forced_cols * len <= 8192 ? forced_cols : (len <= 4096 && !narrow ? 2 : 1);
  const int groups = (nc + cols - 1) / cols;
  const int gpx = (groups + kNumXcd - 1) / kNumXcd;
  const size_t lds = static_cast<size_t>(cols) * lds_elems(len) * sizeof(double2);
  const int threads = forced_threads > 0 ? forced_threads : (narrow && cols == 1 ? 256 : fft_threads(cols * len / 2));
  if (inverse) {
    if (int rc = allow_lds(fft_cols_c2c<true>, lds)) return rc;
    hipLaunchKernelGGL(fft_cols_c2c<true>, dim3(gpx * kNumXcd), dim3(threads), lds, stream, in, d, nc, cols,
                       scale, out, groups, gpx, weights);
  } else {
    if (int rc = allow_lds(fft_cols_c2c<false>, lds)) return rc;
    hipLaunchKernelGGL(fft_cols_c2c<false>, dim3(gpx * kNumXcd), dim3(threads), lds, stream, in, d, nc, cols,
                       scale, out, groups, gpx, weights);
  }
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

}  // namespace
}  // namespace psh

using psh::fail;

// numpy.fft.rfft2 of a real (m, n) float64 array -> (m, n/2+1) complex128
extern "C" int psh_fft_rfft2_dev(const double *in_dev, int m, int n, void *out_dev) {
  PSH_REQUIRE_INIT();
  if (!in_dev || !out_dev) return fail(PSH_EINVAL, "rfft2: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::Dft rows, cols;
  if (int rc = psh::check_shape("rfft2", m, n, &rows, &cols)) return rc;
  const int len = 1 << rows.logm;
  const size_t lds = static_cast<size_t>(psh::lds_elems(len)) * sizeof(double2);
  if (int rc = psh::allow_lds(psh::fft_rows_r2c, lds)) return rc;
  double2 *out = static_cast<double2 *>(out_dev);
  hipLaunchKernelGGL(psh::fft_rows_r2c, dim3((m + 1) / 2), dim3(psh::fft_threads(len)), lds, c.stream, in_dev, m, rows, out);
  PSH_HIP(hipGetLastError());
  return psh::launch_cols(false, out, cols, n / 2 + 1, 1.0, out, c.stream);
}

namespace psh {
void fft_release() {  // psh_shutdown: the tables belong to the device that is being released
  for (auto &kv : twiddle_cache()) (void)hipFree(kv.second);
  twiddle_cache().clear();
  for (auto &kv : chirp_cache()) {
    (void)hipFree(kv.second.chirp);
    (void)hipFree(kv.second.filter);
  }
  chirp_cache().clear();
}
}  // namespace psh

// irfft2(spectrum * weights) -> real (m, n); weights (m, n/2+1) float64 or nullptr; the spectrum is
// left untouched (the column pass writes into `scratch`, (m, n/2+1) complex128).  Lock held.
namespace psh {
int fft_irfft2_weighted(const void *spec_dev, const double *weights_dev, int m, int n, double *out_dev,
                        void *scratch_dev, unsigned long long *min_key_dev) {
  Dft rows, cols;
  if (int rc = check_shape("irfft2", m, n, &rows, &cols)) return rc;
  Context &c = ctx();
  const int nc = n / 2 + 1;
  if (int rc = launch_cols(true, static_cast<const double2 *>(spec_dev), cols, nc, 1.0,
                           static_cast<double2 *>(scratch_dev), c.stream, weights_dev))
    return rc;
  const int len = 1 << rows.logm;
  const size_t lds = static_cast<size_t>(lds_elems(len)) * sizeof(double2);
  if (int rc = allow_lds(fft_rows_c2r, lds)) return rc;
  hipLaunchKernelGGL(fft_rows_c2r, dim3((m + 1) / 2), dim3(fft_threads(len)), lds, c.stream,
                     static_cast<const double2 *>(scratch_dev), m, rows,
                     1.0 / (static_cast<double>(m) * static_cast<double>(n)), out_dev, min_key_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}
bool fft_shape_supported(int m, int n) {
  auto side = [](int v) {
    const int l = ilog2_exact(v);
    return (l >= 1 && l <= kFftMaxLog) || (v >= 2 && v <= kBluesteinMax);
  };
  return side(m) && side(n);
}
}  // namespace psh

// numpy.fft.irfft2(X, s=(m, n)) of an (m, n/2+1) complex128 array -> real (m, n) float64; the input is
// left untouched (the column pass writes into a scratch block)
extern "C" int psh_fft_irfft2_dev(const void *in_dev, int m, int n, double *out_dev) {
  PSH_REQUIRE_INIT();
  if (!in_dev || !out_dev) return fail(PSH_EINVAL, "irfft2: NULL pointer");
  if (!psh::fft_shape_supported(m, n))
    return fail(PSH_EUNSUPPORTED, "irfft2: (%d,%d) - sides: powers of two up to 8192 or any length up to 4096", m, n);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  void *scratch = nullptr;
  if (int rc = psh_malloc(&scratch, static_cast<size_t>(m) * (n / 2 + 1) * sizeof(double2))) return rc;
  const int rc = psh::fft_irfft2_weighted(in_dev, nullptr, m, n, out_dev, scratch);
  (void)psh_free(scratch);  // stream-ordered
  return rc;
}

// ... and np.min of the result as the order-preserving key psh_steps_mask_dev reads (what psh_field_min_key_dev
// computes in a sweep of its own): *min_key_dev is set to the identity first, the row pass folds its outputs in
extern "C" int psh_fft_irfft2_min_dev(const void *in_dev, int m, int n, double *out_dev, unsigned long long *min_key_dev) {
  PSH_REQUIRE_INIT();
  if (!in_dev || !out_dev || !min_key_dev) return fail(PSH_EINVAL, "irfft2_min: NULL pointer");
  if (!psh::fft_shape_supported(m, n))
    return fail(PSH_EUNSUPPORTED, "irfft2: (%d,%d) - sides: powers of two up to 8192 or any length up to 4096", m, n);
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  void *scratch = nullptr;
  if (int rc = psh_malloc(&scratch, static_cast<size_t>(m) * (n / 2 + 1) * sizeof(double2))) return rc;
  PSH_HIP(hipMemsetAsync(min_key_dev, 0xff, sizeof(unsigned long long), c.stream));
  const int rc = psh::fft_irfft2_weighted(in_dev, nullptr, m, n, out_dev, scratch, min_key_dev);
  (void)psh_free(scratch);  // stream-ordered
  return rc;
}

// numpy.fft.fft2 / ifft2 of an (m, n) complex128 array (in_dev == out_dev allowed)
extern "C" int psh_fft_c2c2_dev(const void *in_dev, int m, int n, int inverse, void *out_dev) {
  PSH_REQUIRE_INIT();
  if (!in_dev || !out_dev) return fail(PSH_EINVAL, "fft2: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  psh::Dft rows, cols;
  if (int rc = psh::check_shape("fft2", m, n, &rows, &cols)) return rc;
  const int len = 1 << rows.logm;
  const size_t lds = static_cast<size_t>(psh::lds_elems(len)) * sizeof(double2);
  const double2 *in = static_cast<const double2 *>(in_dev);
  double2 *out = static_cast<double2 *>(out_dev);
  if (inverse) {
    if (int rc = psh::allow_lds(psh::fft_rows_c2c<true>, lds)) return rc;
    hipLaunchKernelGGL(psh::fft_rows_c2c<true>, dim3(m), dim3(psh::fft_threads(len)), lds, c.stream, in, rows, out);
  } else {
    if (int rc = psh::allow_lds(psh::fft_rows_c2c<false>, lds)) return rc;
    hipLaunchKernelGGL(psh::fft_rows_c2c<false>, dim3(m), dim3(psh::fft_threads(len)), lds, c.stream, in, rows, out);
  }
  PSH_HIP(hipGetLastError());
  const double scale = inverse ? 1.0 / (static_cast<double>(m) * static_cast<double>(n)) : 1.0;
  return psh::launch_cols(inverse != 0, out, cols, n, scale, out, c.stream);
}
