// device_prims.hip — rocPRIM-backed sort / scan used by the voxelisation stage (see device_prims.h).
#include "device_prims.h"

#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace dmsa {

size_t sort_pairs_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 64);
    size_t b32 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b32, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, n, 0, 32);
    const size_t own = sort_pairs_u32_workspace_bytes(n);
    bytes = bytes > b32 ? bytes : b32;
    return bytes > own ? bytes : own;
}
size_t scan_temp_bytes(size_t n) {
    size_t a = 0, b = 0;
    (void)rocprim::inclusive_scan(nullptr, a, (const int32_t*)nullptr, (int32_t*)nullptr, n, rocprim::plus<int32_t>());
    (void)rocprim::exclusive_scan(nullptr, b, (const int32_t*)nullptr, (int32_t*)nullptr, int32_t(0), n, rocprim::plus<int32_t>());
    return a > b ? a : b;
}
hipError_t sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                              uint32_t* vals_out, size_t n, unsigned end_bit, hipStream_t stream) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream);
}
hipError_t sort_pairs_u32_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                              uint32_t* vals_out, size_t n, unsigned end_bit, hipStream_t stream, bool library_sort, bool header_zeroed) {
    if (!library_sort && end_bit <= 32) return sort_pairs_u32_onesweep(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, stream, header_zeroed ? 1 : 0);
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream);
}
hipError_t inclusive_scan_i32(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t stream) {
    return rocprim::inclusive_scan(temp, temp_bytes, in, out, n, rocprim::plus<int32_t>(), stream);
}
hipError_t exclusive_scan_i32(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t stream) {
    return rocprim::exclusive_scan(temp, temp_bytes, in, out, int32_t(0), n, rocprim::plus<int32_t>(), stream);
}

}  // namespace dmsa
