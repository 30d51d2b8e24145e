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

}  // namespace
}  // namespace psh

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
