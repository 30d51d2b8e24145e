// Known-traffic streaming copies used to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reads wide
// coalesced streams by 2x; other widths must be calibrated in the access
// pattern of the kernel under study).  Measurement aids only: built into tools/calib/libpsh_calib.so
// (python -m tools.calib.build), NOT into the product library.  Self-contained: launches go to the
// null stream of the current device and synchronise; device pointers come from psh_malloc.
#include <hip/hip_runtime.h>

#include <cstddef>

namespace psh {
namespace {

__global__ __launch_bounds__(256) void calib_copy_dword(float *__restrict__ dst,
                                                        const float *__restrict__ src, size_t n) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = src[i];
}

__global__ __launch_bounds__(256) void calib_copy_dwordx4(float4 *__restrict__ dst,
                                                          const float4 *__restrict__ src,
                                                          size_t n4) {
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride)
    dst[i] = src[i];
}

// L1-resident gather throughput by access width and alignment (how many CU clocks a wave64
// buffer load of 1 / 2 / 4 dwords per lane costs the vector memory pipeline): every wave reads
// the same eight image rows over and over, lane i at column x0 + i * W + shift.
template <int W>
__global__ __launch_bounds__(256) void calib_gather(const float *__restrict__ src, float *__restrict__ sink,
                                                    int pitch_bytes, int shift, int iters, int row_mask, int row_step,
                                                    int first_lane, int lane_stride) {
  typedef float vec __attribute__((ext_vector_type(W)));
  __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, pitch_bytes * (row_mask + 1), 0x00020000);
  const int lane = threadIdx.x & 63;
  const unsigned off = static_cast<unsigned>((blockIdx.x & 1) * 2048 + lane * lane_stride + shift) * 4u;
  // every wave starts somewhere else in the row cycle: no L1 sharing between waves beyond 8 rows
  const int phase = row_step == 1 ? 0 : static_cast<int>((blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 * 37);
  float acc = 0.f;
  if (lane < first_lane) return;  // exec-masked loads: what does a load for a few lanes cost?
  for (int it = 0; it < iters; ++it) {
    vec v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (W == 1)
        v[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, static_cast<int>(off), ((k + it * row_step + phase) & row_mask) * pitch_bytes, 0));
      else if constexpr (W == 2)
        v[k] = __builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b64(r, static_cast<int>(off), ((k + it * row_step + phase) & row_mask) * pitch_bytes, 0));
      else
        v[k] = __builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b128(r, static_cast<int>(off), ((k + it * row_step + phase) & row_mask) * pitch_bytes, 0));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int j = 0; j < W; ++j) acc += v[k][j];
    }
    asm volatile("" ::: "memory");
  }
  if (acc == 123.456f) sink[threadIdx.x] = acc;
}

// which lane does each DPP control deliver?  out[c * 64 + lane] = lane id received (-1: none)
__global__ __launch_bounds__(64) void calib_dpp(int *__restrict__ out) {
  const int lane = threadIdx.x;
  out[0 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  out[1 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
  out[2 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
  out[3 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
  out[4 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
  out[5 * 64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
}


// VALU issue cost of single instructions (wave64, gfx950): `waves` waves per SIMD each run `iters`
// rounds of 8 independent instances of one instruction; the cycles of a round / 8 are reported by
// the host wrapper as (kernel time x clock) / instructions per SIMD.
//  0 v_fma_f32   1 v_pk_fma_f32   2 v_sqrt_f32   3 v_rsq_f32   4 v_add_f64   5 v_cvt_f64_f32
//  6 v_fma_f64   7 v_mov_b32 dpp row_shr:1   8 v_rcp_f32   9 v_cvt_f32_f64   10 v_pk_add_f32
//  11 v_pk_mul_f32  12 v_add_f32 dpp  13 v_mul_f64  14 v_add_u32  15 ds_bpermute (shfl)
template <int OP>
__global__ __launch_bounds__(256) void calib_valu(float *__restrict__ sink, int iters) {
  float a[8];
  double d[8];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a[k] = 1.0f + 0.001f * static_cast<float>(threadIdx.x + k);
    d[k] = 1.0 + 0.001 * static_cast<double>(threadIdx.x + k);
    p[k] = v2f{a[k], a[k] + 0.5f};
  }
  const float c = 0.999f, e = 1e-6f;
  const double dc = 1e-9;
  const v2f pc = {0.999f, 0.998f}, pe = {1e-6f, 2e-6f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c), "v"(e));
      if constexpr (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pc), "v"(pe));
      if constexpr (OP == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]));
      if constexpr (OP == 3) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]));
      if constexpr (OP == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
      if constexpr (OP == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]));
      if constexpr (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[k]) : "v"(dc));
      if constexpr (OP == 7) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
      if constexpr (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
      if constexpr (OP == 9) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]));
      if constexpr (OP == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pe));
      if constexpr (OP == 11) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc));
      if constexpr (OP == 12) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[k]));
      if constexpr (OP == 13) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
      if constexpr (OP == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
      if constexpr (OP == 15) a[k] = __shfl_xor(a[k], 1);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += a[k] + static_cast<float>(d[k]) + p[k].x + p[k].y;
  if (s == 123.456f) sink[threadIdx.x] = s;
}

}  // namespace
}  // namespace psh

#define CALIB_HIP(expr)            \
  do {                             \
    hipError_t _e = (expr);        \
    if (_e != hipSuccess) return static_cast<int>(_e); \
  } while (0)

extern "C" int calib_dpp(int *out_dev) {
  hipLaunchKernelGGL(psh::calib_dpp, dim3(1), dim3(64), 0, nullptr, out_dev);
  CALIB_HIP(hipGetLastError());
  CALIB_HIP(hipDeviceSynchronize());
  return 0;
}

// lane i reads `width` dwords at column x0 + i * lane_stride + shift (lane_stride < width: the
// lanes' footprints overlap, as in the packed semi-Lagrangian gathers); returns milliseconds in *ms
extern "C" int calib_gather(const float *src_dev, float *sink_dev, int pitch_bytes, int width, int lane_stride,
                            int shift, int iters, int blocks_per_cu, int n_rows, int active_lanes, float *ms) {
  if (!src_dev || !sink_dev || n_rows < 8 || (n_rows & (n_rows - 1))) return -1;
  int dev = 0, cus = 0;
  CALIB_HIP(hipGetDevice(&dev));
  CALIB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const dim3 grid(cus * blocks_per_cu), block(256);
  hipEvent_t e0, e1;
  CALIB_HIP(hipEventCreate(&e0));
  CALIB_HIP(hipEventCreate(&e1));
  CALIB_HIP(hipEventRecord(e0, nullptr));
  const int row_step = n_rows > 8 ? 8 : 1, first = 64 - active_lanes;
  if (width == 1)
    hipLaunchKernelGGL(psh::calib_gather<1>, grid, block, 0, nullptr, src_dev, sink_dev, pitch_bytes, shift, iters,
                       n_rows - 1, row_step, first, lane_stride);
  else if (width == 2)
    hipLaunchKernelGGL(psh::calib_gather<2>, grid, block, 0, nullptr, src_dev, sink_dev, pitch_bytes, shift, iters,
                       n_rows - 1, row_step, first, lane_stride);
  else if (width == 4)
    hipLaunchKernelGGL(psh::calib_gather<4>, grid, block, 0, nullptr, src_dev, sink_dev, pitch_bytes, shift, iters,
                       n_rows - 1, row_step, first, lane_stride);
  else
    return -1;
  CALIB_HIP(hipGetLastError());
  CALIB_HIP(hipEventRecord(e1, nullptr));
  CALIB_HIP(hipEventSynchronize(e1));
  if (ms) CALIB_HIP(hipEventElapsedTime(ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}


// one instruction kind, `blocks_per_cu` 256-thread workgroups per CU (= waves per SIMD), `iters` rounds of 8
// instructions per wave; *ms = kernel time (events on the null stream)
extern "C" int calib_valu(float *sink_dev, int op, int blocks_per_cu, int iters, float *ms) {
  int dev = 0, cus = 0;
  CALIB_HIP(hipGetDevice(&dev));
  CALIB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const dim3 grid(cus * blocks_per_cu), block(256);
  hipEvent_t e0, e1;
  CALIB_HIP(hipEventCreate(&e0));
  CALIB_HIP(hipEventCreate(&e1));
  CALIB_HIP(hipEventRecord(e0, nullptr));
#define CALIB_V(OP) \
  case OP: hipLaunchKernelGGL(psh::calib_valu<OP>, grid, block, 0, nullptr, sink_dev, iters); break;
  switch (op) {
    CALIB_V(0) CALIB_V(1) CALIB_V(2) CALIB_V(3) CALIB_V(4) CALIB_V(5) CALIB_V(6) CALIB_V(7)
    CALIB_V(8) CALIB_V(9) CALIB_V(10) CALIB_V(11) CALIB_V(12) CALIB_V(13) CALIB_V(14) CALIB_V(15)
    default: return -1;
  }
#undef CALIB_V
  CALIB_HIP(hipGetLastError());
  CALIB_HIP(hipEventRecord(e1, nullptr));
  CALIB_HIP(hipEventSynchronize(e1));
  if (ms) CALIB_HIP(hipEventElapsedTime(ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

extern "C" int calib_copy(float *dst_dev, const float *src_dev, size_t nfloats, int vec_width) {
  if (!dst_dev || !src_dev) return -1;
  int dev = 0, cus = 0;
  CALIB_HIP(hipGetDevice(&dev));
  CALIB_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = cus * 8;
  if (vec_width == 4) {
    if (nfloats % 4) return -1;
    hipLaunchKernelGGL(psh::calib_copy_dwordx4, dim3(grid), dim3(256), 0, nullptr, reinterpret_cast<float4 *>(dst_dev),
                       reinterpret_cast<const float4 *>(src_dev), nfloats / 4);
  } else if (vec_width == 1) {
    hipLaunchKernelGGL(psh::calib_copy_dword, dim3(grid), dim3(256), 0, nullptr, dst_dev, src_dev, nfloats);
  } else {
    return -1;
  }
  CALIB_HIP(hipGetLastError());
  CALIB_HIP(hipDeviceSynchronize());
  return 0;
}
