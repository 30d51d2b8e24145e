// Known-traffic streaming copies used to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md "HBM": FETCH_SIZE under-reads wide
// coalesced streams by 2x; other widths must be calibrated in the access
// pattern of the kernel under study).  Not part of the hot path.
#include "common.h"

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
                                                    int first_lane) {
  typedef float vec __attribute__((ext_vector_type(W)));
  __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, pitch_bytes * (row_mask + 1), 0x00020000);
  const int lane = threadIdx.x & 63;
  const unsigned off = static_cast<unsigned>((blockIdx.x & 1) * 2048 + lane * W + shift) * 4u;
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

}  // namespace
}  // namespace psh

extern "C" int psh_calib_dpp(int *out_dev) {
  PSH_REQUIRE_INIT();
  if (!out_dev) return psh::fail(PSH_EINVAL, "psh_calib_dpp: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  hipLaunchKernelGGL(psh::calib_dpp, dim3(1), dim3(64), 0, c.stream, out_dev);
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_calib_gather(const float *src_dev, float *sink_dev, int pitch_bytes, int width, int shift,
                                int iters, int blocks_per_cu, int n_rows, int active_lanes) {
  PSH_REQUIRE_INIT();
  if (!src_dev || !sink_dev) return psh::fail(PSH_EINVAL, "psh_calib_gather: NULL pointer");
  if (n_rows < 8 || (n_rows & (n_rows - 1))) return psh::fail(PSH_EINVAL, "psh_calib_gather: n_rows must be a power of two >= 8");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const dim3 grid(c.cu_count * blocks_per_cu), block(256);
  if (width == 1)
    hipLaunchKernelGGL(psh::calib_gather<1>, grid, block, 0, c.stream, src_dev, sink_dev, pitch_bytes, shift, iters, n_rows - 1, n_rows > 8 ? 8 : 1,
                       64 - active_lanes);
  else if (width == 2)
    hipLaunchKernelGGL(psh::calib_gather<2>, grid, block, 0, c.stream, src_dev, sink_dev, pitch_bytes, shift, iters, n_rows - 1, n_rows > 8 ? 8 : 1,
                       64 - active_lanes);
  else if (width == 4)
    hipLaunchKernelGGL(psh::calib_gather<4>, grid, block, 0, c.stream, src_dev, sink_dev, pitch_bytes, shift, iters, n_rows - 1, n_rows > 8 ? 8 : 1,
                       64 - active_lanes);
  else
    return psh::fail(PSH_EINVAL, "psh_calib_gather: width must be 1, 2 or 4");
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}

extern "C" int psh_calib_copy(float *dst_dev, const float *src_dev, size_t nfloats, int vec_width) {
  PSH_REQUIRE_INIT();
  if (!dst_dev || !src_dev) return psh::fail(PSH_EINVAL, "psh_calib_copy: NULL pointer");
  psh::Context &c = psh::ctx();
  std::lock_guard<std::recursive_mutex> lock(c.mu);
  PSH_HIP(hipSetDevice(c.device));
  const int grid = c.cu_count * 8;
  if (vec_width == 4) {
    if (nfloats % 4) return psh::fail(PSH_EINVAL, "psh_calib_copy: nfloats must be a multiple of 4");
    hipLaunchKernelGGL(psh::calib_copy_dwordx4, dim3(grid), dim3(256), 0, c.stream,
                       reinterpret_cast<float4 *>(dst_dev),
                       reinterpret_cast<const float4 *>(src_dev), nfloats / 4);
  } else if (vec_width == 1) {
    hipLaunchKernelGGL(psh::calib_copy_dword, dim3(grid), dim3(256), 0, c.stream, dst_dev, src_dev,
                       nfloats);
  } else {
    return psh::fail(PSH_EINVAL, "psh_calib_copy: vec_width must be 1 or 4");
  }
  PSH_HIP(hipGetLastError());
  return PSH_OK;
}
