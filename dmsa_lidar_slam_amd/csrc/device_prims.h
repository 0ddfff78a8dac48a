// device_prims.h — the device-wide primitives of the voxelisation (radix sort of pairs, prefix scans), all hand-written: radix_sort.hip.
// (Rounds 1-4 kept rocPRIM behind the 64-bit sort and the scans of the rows before the hot path; round 5 removed the library.)
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace dmsa {
// temp-storage queries (bytes) for n elements
size_t sort_pairs_temp_bytes(size_t n);
size_t scan_temp_bytes(size_t n);
// stable LSD radix sort of (key u64, value u32) pairs on bits [0, end_bit)
hipError_t sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in,
                              uint32_t* vals_out, size_t n, unsigned end_bit, hipStream_t stream);
hipError_t sort_pairs_u32_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                              uint32_t* vals_out, size_t n, unsigned end_bit, hipStream_t stream, bool header_zeroed = false);
// radix_sort.hip: the hand-written onesweep sort behind sort_pairs_u32_u32
size_t sort_pairs_u32_workspace_bytes(size_t n);
hipError_t sort_pairs_u32_onesweep(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                                   size_t n, unsigned end_bit, hipStream_t stream, int prepared = 0 /* 1: header already zeroed, 2: header + look-back state + histograms done */);
void sort_set_items_override(int items);  // debug switch sort_items: pairs per thread of a sort tile (2, 4, 8, 16; 0 = by size), process-wide
hipError_t inclusive_scan_i32(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t stream);
hipError_t exclusive_scan_i32(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t stream);
}  // namespace dmsa
