// radix_sort.hip — hand-written stable LSD radix sort of (u32 key, u32 value) pairs for gfx950 (voxelisation stage).
//
// The voxelisation sorts 1.5 M (leaf code, point index) pairs per resolution and iteration on 17..24 key bits.  A library sort
// spends more dispatches on bookkeeping (memset fills of its histograms and look-back state, a separate scan) than on the three
// scatter passes, and every dispatch of a dependent chain costs a few microseconds of launch gap.  This one is four or five dispatches:
//
//   k_sort_hist   one read of the keys: the 256-bin histograms of ALL digits (LDS atomics, one global atomic per bin and
//                 workgroup); its workgroups also clear the look-back state of the passes
//   k_sort_pass   x ceil(bits / 8): "onesweep" -- a tile of 8192 pairs per workgroup, ranks by wave-level digit matching (eight
//                 ballots per key, one LDS counter per (wave, digit)), tile prefix by decoupled look-back (thread d follows digit d
//                 through the earlier tiles' published counts), pairs reordered in LDS so that every digit leaves the tile as one
//                 contiguous run.  Tiles are handed out by an atomic ticket, so a tile only ever waits for tiles that started earlier.
//
// Stable (elements of a tile are ranked in memory order, tiles in ticket order), deterministic, bit-identical to any other stable
// sort -- the parity tests compare the resulting leaf order with the oracle's.
#include "device_prims.h"
#include "radix_sort_dev.h"

#include <cstdint>

namespace dmsa {

namespace {

constexpr int kBins = kSortBins;
#ifndef DMSA_SORT_THREADS
#define DMSA_SORT_THREADS 512
#endif
#ifndef DMSA_SORT_LOOKBACK
#define DMSA_SORT_LOOKBACK 16
#endif
constexpr int kSortThreads = DMSA_SORT_THREADS, kSortWaves = kSortThreads / 64, kItems = 16, kTile = kSortThreads * kItems;  // pairs per tile
constexpr int kLook = DMSA_SORT_LOOKBACK;  // predecessors inspected per round trip of the look-back
static_assert(kSortThreads >= kBins && kSortThreads % 64 == 0, "one thread per digit");
constexpr int kHistItems = 32;                                                                                   // 8192 keys per histogram workgroup
constexpr int kMaxPasses = kSortMaxPasses;
constexpr uint32_t kFlagPartial = 1u << 30, kFlagPrefix = 2u << 30, kValMask = (1u << 30) - 1;

__global__ __launch_bounds__(256) void k_sort_zero(SortHeader* h) {
    uint32_t* w = reinterpret_cast<uint32_t*>(h);
    for (unsigned i = threadIdx.x; i < sizeof(SortHeader) / 4; i += 256) w[i] = 0;
}

__global__ __launch_bounds__(256) void k_sort_hist(const uint32_t* __restrict__ keys, size_t n, int passes, uint32_t last_mask, SortHeader* __restrict__ h,
                                                   uint32_t* __restrict__ tile_state, size_t state_words) {
    __shared__ uint32_t s_h[kMaxPasses][kBins];
    const int tid = threadIdx.x;
    for (int p = 0; p < kMaxPasses; ++p) s_h[p][tid] = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < state_words; i += (size_t)gridDim.x * 256) tile_state[i] = 0;
    __syncthreads();
    // Leaf codes of neighbouring points share their upper digits: counting key by key would send all 64 lanes of a wave to the same LDS
    // counter.  Every thread walks its own 32 consecutive keys and counts RUNS of equal digits in registers; one LDS atomic per run.
    const size_t base = (size_t)blockIdx.x * (256 * kHistItems) + (size_t)tid * kHistItems;
    uint32_t prev[kMaxPasses] = {0, 0, 0, 0}, run[kMaxPasses] = {0, 0, 0, 0};
    const bool aligned16 = (reinterpret_cast<uintptr_t>(keys) & 15u) == 0;  // a view into a larger array (level 1 behind level 0) may not be
    if (base < n) {
        const size_t m = n - base < (size_t)kHistItems ? n - base : (size_t)kHistItems;
        for (size_t k4 = 0; k4 < m; k4 += 4) {
            uint32_t kk[4];
            if (k4 + 4 <= m && aligned16) {
                const uint4 v = *reinterpret_cast<const uint4*>(keys + base + k4);  // base is a multiple of 32 keys
                kk[0] = v.x, kk[1] = v.y, kk[2] = v.z, kk[3] = v.w;
            } else {
                for (int u = 0; u < 4; ++u) kk[u] = k4 + u < m ? keys[base + k4 + u] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k4 + u < m) {
#pragma unroll
                    for (int p = 0; p < kMaxPasses; ++p) {
                        if (p < passes) {
                            const uint32_t d = sort_digit(kk[u], p, passes, last_mask);
                            if (d == prev[p]) {
                                ++run[p];
                            } else {
                                if (run[p]) atomicAdd(&s_h[p][prev[p]], run[p]);
                                prev[p] = d, run[p] = 1;
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int p = 0; p < kMaxPasses; ++p)
            if (p < passes && run[p]) atomicAdd(&s_h[p][prev[p]], run[p]);
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const uint32_t v = s_h[p][tid];
        if (v) atomicAdd(&h->hist[p][tid], v);
    }
}

// exclusive prefix sum over the 256 threads of the workgroup (value per thread), via wave scans + one LDS exchange
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* s_wave /* [kSortWaves] */, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
        const uint32_t t = s_wave[w];
        if (w < wave) before += t;
        all += t;
    }
    __syncthreads();
    total = all;
    return before + incl - v;
}

__global__ __launch_bounds__(kSortThreads) void k_sort_pass(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int pass, int shift, uint32_t digit_mask,
                                                            SortHeader* __restrict__ h, uint32_t* __restrict__ tile_state /* [tiles][256] of this pass */) {
    __shared__ uint32_t s_cnt[kSortWaves][kBins];  // per wave and digit: running count while ranking, then the wave's offset inside the digit
    __shared__ uint32_t s_base[kBins];              // where the digit's run of this tile starts in the output
    __shared__ uint32_t s_excl[kBins];              // where it starts inside the tile
    __shared__ uint32_t s_keys[kTile], s_vals[kTile];
    __shared__ uint32_t s_wave[kSortWaves];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_tile = atomicAdd(&h->ticket[pass], 1u);
    for (int i = tid; i < kSortWaves * kBins; i += kSortThreads) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    const size_t base = (size_t)tile * kTile;

    // ---- load: wave w owns rows [16 w, 16 w + 16) of 64 consecutive pairs; memory order = (row, lane) ----
    uint32_t key[kItems], val[kItems], rank[kItems];
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const size_t i = base + (size_t)(wave * kItems + k) * 64 + lane;
        const bool valid = i < n;
        key[k] = valid ? keys_in[i] : 0xffffffffu;
        val[k] = valid ? vals_in[i] : 0u;
    }
    // ---- rank inside the wave, row by row ----
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const size_t i = base + (size_t)(wave * kItems + k) * 64 + lane;
        const bool valid = i < n;
        const uint32_t d = (key[k] >> shift) & digit_mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t vote = __ballot(bit);
            peers &= bit ? vote : ~vote;
        }
        const uint32_t below = (uint32_t)__popcll(peers & lt_mask);
        const uint32_t old = s_cnt[wave][d];
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0) s_cnt[wave][d] = old + (uint32_t)__popcll(peers);  // one writer per digit and row
        __builtin_amdgcn_wave_barrier();
        rank[k] = old + below;
    }
    __syncthreads();
    // ---- digit d = tid: offsets of the waves inside the digit, tile count, position of the digit inside the tile ----
    const bool digit_thread = tid < kBins;
    uint32_t count = 0;
    if (digit_thread) {
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t c = s_cnt[w][tid];
            s_cnt[w][tid] = count;
            count += c;
        }
    }
    uint32_t tile_total, bins_total;
    const uint32_t local_excl = block_exclusive_scan(count, s_wave, tile_total);
    const uint32_t bin_base = block_exclusive_scan(digit_thread ? h->hist[pass][tid] : 0u, s_wave, bins_total);
    // ---- decoupled look-back over the tiles with a smaller ticket ----
    uint32_t excl = 0;
    if (digit_thread) {
        uint32_t* st = tile_state + (size_t)tile * kBins + tid;
        if (tile == 0) {
            __hip_atomic_store(st, kFlagPrefix | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(st, kFlagPartial | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // kLook predecessors per round trip: the loads are independent, their results are consumed in order
            int64_t t = (int64_t)tile - 1;
            bool done = false;
            while (!done) {
                uint32_t v[kLook];
#pragma unroll
                for (int u = 0; u < kLook; ++u)
                    v[u] = t - u >= 0 ? __hip_atomic_load(tile_state + (size_t)(t - u) * kBins + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (kFlagPrefix | 0u);
                int used = 0;
#pragma unroll
                for (int u = 0; u < kLook; ++u) {
                    if (!done && used == u) {
                        const uint32_t f = v[u] >> 30;
                        if (f != 0u) {
                            excl += v[u] & kValMask;
                            ++used;
                            done = f == 2u;
                        }
                    }
                }
                t -= used;
            }
            __hip_atomic_store(st, kFlagPrefix | (excl + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_base[tid] = bin_base + excl;
        s_excl[tid] = local_excl;
    }
    __syncthreads();
    // ---- reorder inside the tile: every digit becomes one contiguous run ----
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
        const size_t i = base + (size_t)(wave * kItems + k) * 64 + lane;
        if (i < n) {
            const uint32_t d = (key[k] >> shift) & digit_mask;
            const uint32_t lpos = s_excl[d] + s_cnt[wave][d] + rank[k];
            s_keys[lpos] = key[k], s_vals[lpos] = val[k];
        }
    }
    __syncthreads();
    const uint32_t in_tile = (uint32_t)(n - base < (size_t)kTile ? n - base : (size_t)kTile);
#pragma unroll 4
    for (uint32_t i = tid; i < in_tile; i += kSortThreads) {
        const uint32_t kk = s_keys[i];
        const uint32_t d = (kk >> shift) & digit_mask;
        const size_t pos = (size_t)s_base[d] + (i - s_excl[d]);
        keys_out[pos] = kk, vals_out[pos] = s_vals[i];
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

size_t sort_pairs_u32_workspace_bytes(size_t n) {
    const size_t tiles = (n + kTile - 1) / kTile;
    return align_up(sizeof(SortHeader), 256) + align_up(tiles * kBins * kMaxPasses * 4, 256) + 2 * align_up(n * 4, 256);
}

SortPlan sort_pairs_u32_plan(void* temp, size_t n, unsigned end_bit) {
    const size_t tiles = (n + kTile - 1) / kTile;
    const int passes = end_bit == 0 ? 1 : (int)((end_bit + 7) / 8);
    char* w = static_cast<char*>(temp);
    SortPlan p;
    p.header = reinterpret_cast<SortHeader*>(w);
    p.tile_state = reinterpret_cast<uint32_t*>(w + align_up(sizeof(SortHeader), 256));
    p.passes = passes;
    // stable on the key bits [0, end_bit): whatever lies above takes no part (rocPRIM's begin_bit / end_bit semantics)
    const unsigned top = end_bit == 0 ? 0u : end_bit - 8u * (unsigned)(passes - 1);
    p.last_mask = top >= 8u ? 255u : ((1u << top) - 1u);
    p.state_words = tiles * kBins * (size_t)(passes < kMaxPasses ? passes : kMaxPasses);
    return p;
}

hipError_t sort_pairs_u32_onesweep(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out,
                                   size_t n, unsigned end_bit, hipStream_t stream, int prepared) {
    if (n == 0) return hipSuccess;
    if (n >= (size_t)kValMask || temp_bytes < sort_pairs_u32_workspace_bytes(n)) return hipErrorInvalidValue;
    const SortPlan plan = sort_pairs_u32_plan(temp, n, end_bit);
    const int passes = plan.passes;
    if (passes > kMaxPasses) return hipErrorInvalidValue;
    const size_t tiles = (n + kTile - 1) / kTile;
    char* w = static_cast<char*>(temp) + align_up(sizeof(SortHeader), 256) + align_up(tiles * kBins * kMaxPasses * 4, 256);
    uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(w);
    w += align_up(n * 4, 256);
    uint32_t* vals_tmp = reinterpret_cast<uint32_t*>(w);
    SortHeader* h = plan.header;
    uint32_t* state = plan.tile_state;
    if (prepared < 2) {  // 2: the producer of the keys cleared the header and the look-back words and counted the digits
        if (prepared < 1) hipLaunchKernelGGL(k_sort_zero, dim3(1), dim3(256), 0, stream, h);  // 1: an earlier kernel of the stream cleared the header
        const unsigned hist_blocks = (unsigned)((n + 256 * kHistItems - 1) / (256 * kHistItems));
        hipLaunchKernelGGL(k_sort_hist, dim3(hist_blocks), dim3(256), 0, stream, keys_in, n, passes, plan.last_mask, h, state, plan.state_words);
    }
    // ping-pong so that the last pass writes the caller's output arrays
    const uint32_t* kin = keys_in;
    const uint32_t* vin = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        uint32_t* kout = to_out ? keys_out : keys_tmp;
        uint32_t* vout = to_out ? vals_out : vals_tmp;
        hipLaunchKernelGGL(k_sort_pass, dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, kin, vin, kout, vout, n, p, 8 * p, p == passes - 1 ? plan.last_mask : 255u, h,
                           state + (size_t)p * tiles * kBins);
        kin = kout, vin = vout;
    }
    return hipGetLastError();
}

}  // namespace dmsa
