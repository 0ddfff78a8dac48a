// radix_sort_dev.h — what a kernel that PRODUCES sort keys needs in order to do the sort's bookkeeping on the side (radix_sort.hip):
// the layout of the sort header and the digit histograms of the keys it writes.  The voxelisation's key kernel uses it, which saves
// the sort its clearing kernel and its histogram pass over the keys (two dispatches of a chain in which every dispatch costs its
// launch gap).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dmsa {

constexpr int kSortBins = 256;
constexpr int kSortMaxPasses = 4;

struct SortHeader {
    uint32_t hist[kSortMaxPasses][kSortBins];  // digit histograms of all passes
    uint32_t ticket[kSortMaxPasses];           // tile tickets of the passes
    uint32_t pad[60];
};

// where the pieces of a sort workspace live (host side; radix_sort.hip)
struct SortPlan {
    SortHeader* header;
    uint32_t* tile_state;  // look-back words of all passes
    size_t state_words;
    int passes;
    uint32_t last_mask;  // digit mask of the LAST pass: key bits at or above end_bit take no part in the sort (255 when end_bit is a multiple of 8)
};
SortPlan sort_pairs_u32_plan(void* temp, size_t n, unsigned end_bit);

#ifdef __HIPCC__
// digit of pass p: the last pass only looks at the key bits below the sort's end bit
__device__ __forceinline__ uint32_t sort_digit(uint32_t key, int p, int passes, uint32_t last_mask) {
    return (key >> (8 * p)) & (p == passes - 1 ? last_mask : 255u);
}
// One key of the calling lane into the workgroup's LDS histograms.  Leaf codes of neighbouring points share their upper digits: when
// all active lanes of the wave hold the same digit, one lane adds the whole count.
__device__ __forceinline__ void sort_hist_add(uint32_t (*s_h)[kSortBins], int passes, uint32_t last_mask, uint32_t key, bool valid) {
    const unsigned long long vm = __ballot(valid);
    if (vm == 0ull) return;
    const int leader = __ffsll((long long)vm) - 1;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int p = 0; p < kSortMaxPasses; ++p) {
        if (p < passes) {
            const uint32_t d = sort_digit(key, p, passes, last_mask);
            const uint32_t d0 = (uint32_t)__shfl((int)d, leader);
            if (__ballot(valid && d != d0) == 0ull) {
                if (lane == leader) atomicAdd(&s_h[p][d0], (uint32_t)__popcll(vm));
            } else if (valid) {
                atomicAdd(&s_h[p][d], 1u);
            }
        }
    }
}
#endif

}  // namespace dmsa
