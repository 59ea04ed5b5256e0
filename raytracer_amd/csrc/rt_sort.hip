// rt_sort.hip -- key/value radix sort of the bounce-ray term list (rocPRIM), kept in its own translation unit.
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "rt_kernels.h"

// temp == nullptr: only *temp_bytes is set.  Sorts n (key, value) pairs by bits [0, end_bit) of the key.
hipError_t rt_sort_terms(void* temp, size_t* temp_bytes, const unsigned long long* keys_in, unsigned long long* keys_out,
                         const RtTermRGB* vals_in, RtTermRGB* vals_out, uint32_t n, unsigned end_bit, hipStream_t stream) {
  return rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, stream);
}

hipError_t rt_sort_u32(void* temp, size_t* temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                       uint32_t* vals_out, uint32_t n, unsigned end_bit, hipStream_t stream) {
  return rocprim::radix_sort_pairs(temp, *temp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, end_bit, stream);
}
