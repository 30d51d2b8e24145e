// Ordering of the Shi-Tomasi corner candidates on the device (rocPRIM radix sort).
//
// goodFeaturesToTrack walks the candidates strongest first, ties by descending address
// (featureselect.cpp greaterThanPtr).  With the key (response bits << 32 | y*n + x) that order
// is a descending unsigned 64-bit sort; only the head of the sorted list ever reaches the host.
// Kept in its own translation unit: rocPRIM's templates dominate the compile time.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.h"

namespace psh {

hipError_t sort_keys_desc(const unsigned long long *in, unsigned long long *out, unsigned int n,
                          void *temp, size_t *temp_bytes, hipStream_t stream) {
  return rocprim::radix_sort_keys_desc(temp, *temp_bytes, in, out, n, 0, 64, stream);
}

}  // namespace psh
