// radix_sort.hip — hand-written stable LSD radix sort of (u32 or u64 key, u32 value) pairs and the prefix scans of int32 arrays for gfx950
// (voxelisation stage and the rows before it).  No vendor library: the path from the points to the Gaussians is this repository's code.
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

#include <algorithm>
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
constexpr int kSortThreads = DMSA_SORT_THREADS, kSortWaves = kSortThreads / 64;
// Pairs per thread of a tile.  16 (8192-pair tiles) is tuned for 10^6 pairs; a window of the reference's everyday size (25 000 points) would be
// FOUR such tiles -- four compute units each ranking 8192 pairs (~8 us of vector issue) while 252 idle.  Small inputs get small tiles: the
// per-tile work shrinks with the tile, the look-back chain stays a round trip or two (kLook predecessors at once).
int g_sort_items_override = 0;  // debug switch sort_items (2, 4, 8, 16; 0 = by size): process-wide, experiments only
inline int sort_items_for(size_t n) {
    if (g_sort_items_override == 2 || g_sort_items_override == 4 || g_sort_items_override == 8 || g_sort_items_override == 16) return g_sort_items_override;
    return n <= (size_t(1) << 16) ? 2 : n <= (size_t(1) << 18) ? 4 : 16;
}
inline size_t sort_tile_for(size_t n) { return (size_t)kSortThreads * sort_items_for(n); }
constexpr int kLook = DMSA_SORT_LOOKBACK;  // predecessors inspected per round trip of the look-back
static_assert(kSortThreads >= kBins && kSortThreads % 64 == 0, "one thread per digit");
__host__ __device__ constexpr int hist_items_for(size_t n) { return n <= (size_t(1) << 16) ? 4 : n <= (size_t(1) << 18) ? 8 : 32; }  // keys per thread of a histogram workgroup (256 threads)
constexpr int kMaxPasses = kSortMaxPasses;
constexpr uint32_t kFlagPartial = 1u << 30, kFlagPrefix = 2u << 30, kValMask = (1u << 30) - 1;

__global__ __launch_bounds__(256) void k_sort_zero(SortHeader* h) {
    uint32_t* w = reinterpret_cast<uint32_t*>(h);
    for (unsigned i = threadIdx.x; i < sizeof(SortHeader) / 4; i += 256) w[i] = 0;
}

template <int kHistItems>
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
                const uint4 v = *reinterpret_cast<const uint4*>(keys + base + k4);  // base is a multiple of four keys
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

template <int kItems>
__global__ __launch_bounds__(kSortThreads) void k_sort_pass(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int pass, int shift, uint32_t digit_mask,
                                                            SortHeader* __restrict__ h, uint32_t* __restrict__ tile_state /* [tiles][256] of this pass */) {
    constexpr int kTile = kSortThreads * kItems;
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

    // ---- load: wave w owns rows [kItems w, kItems w + kItems) of 64 consecutive pairs; memory order = (row, lane) ----
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

void sort_set_items_override(int items) { g_sort_items_override = items; }

size_t sort_pairs_u32_workspace_bytes(size_t n) {
    // sized for the smallest tile any n' <= n may choose (a workspace is allocated once for the largest problem and reused for smaller ones)
    const size_t tiles = (n + (size_t)kSortThreads * 2 - 1) / ((size_t)kSortThreads * 2);
    return align_up(sizeof(SortHeader), 256) + align_up(tiles * kBins * kMaxPasses * 4, 256) + 2 * align_up(n * 4, 256);
}

SortPlan sort_pairs_u32_plan(void* temp, size_t n, unsigned end_bit) {
    const size_t kTile = sort_tile_for(n);
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
    const size_t kTile = sort_tile_for(n);
    const size_t tiles = (n + kTile - 1) / kTile;
    const size_t max_tiles = (n + (size_t)kSortThreads * 2 - 1) / ((size_t)kSortThreads * 2);  // (the layout of sort_pairs_u32_workspace_bytes)
    char* w = static_cast<char*>(temp) + align_up(sizeof(SortHeader), 256) + align_up(max_tiles * kBins * kMaxPasses * 4, 256);
    uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(w);
    w += align_up(n * 4, 256);
    uint32_t* vals_tmp = reinterpret_cast<uint32_t*>(w);
    SortHeader* h = plan.header;
    uint32_t* state = plan.tile_state;
    if (prepared < 2) {  // 2: the producer of the keys cleared the header and the look-back words and counted the digits
        if (prepared < 1) hipLaunchKernelGGL(k_sort_zero, dim3(1), dim3(256), 0, stream, h);  // 1: an earlier kernel of the stream cleared the header
        const int hi = hist_items_for(n);
        const unsigned hist_blocks = (unsigned)((n + 256 * (size_t)hi - 1) / (256 * (size_t)hi));
        if (hi == 4)
            hipLaunchKernelGGL(k_sort_hist<4>, dim3(hist_blocks), dim3(256), 0, stream, keys_in, n, passes, plan.last_mask, h, state, plan.state_words);
        else if (hi == 8)
            hipLaunchKernelGGL(k_sort_hist<8>, dim3(hist_blocks), dim3(256), 0, stream, keys_in, n, passes, plan.last_mask, h, state, plan.state_words);
        else
            hipLaunchKernelGGL(k_sort_hist<32>, dim3(hist_blocks), dim3(256), 0, stream, keys_in, n, passes, plan.last_mask, h, state, plan.state_words);
    }
    // ping-pong so that the last pass writes the caller's output arrays
    const uint32_t* kin = keys_in;
    const uint32_t* vin = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        uint32_t* kout = to_out ? keys_out : keys_tmp;
        uint32_t* vout = to_out ? vals_out : vals_tmp;
        const uint32_t dm = p == passes - 1 ? plan.last_mask : 255u;
        uint32_t* st = state + (size_t)p * tiles * kBins;
        switch (sort_items_for(n)) {
            case 2: hipLaunchKernelGGL(k_sort_pass<2>, dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, kin, vin, kout, vout, n, p, 8 * p, dm, h, st); break;
            case 4: hipLaunchKernelGGL(k_sort_pass<4>, dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, kin, vin, kout, vout, n, p, 8 * p, dm, h, st); break;
            case 8: hipLaunchKernelGGL(k_sort_pass<8>, dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, kin, vin, kout, vout, n, p, 8 * p, dm, h, st); break;
            default: hipLaunchKernelGGL(k_sort_pass<16>, dim3((unsigned)tiles), dim3(kSortThreads), 0, stream, kin, vin, kout, vout, n, p, 8 * p, dm, h, st); break;
        }
        kin = kout, vin = vout;
    }
    return hipGetLastError();
}

// ---- 64-bit keys -----------------------------------------------------------------------------------------------------------------
// Leaf codes wider than 32 bits (octrees deeper than ten levels with the key compression off or exhausted) are rare and not on the timed
// path; they are sorted as two stable 32-bit sorts that carry the POSITION of a pair instead of the pair: low words first, then the high
// words gathered in that order -- an LSD radix sort over eight digits with a gather between the halves.  keys_out / vals_out are gathered once.
namespace {
__global__ __launch_bounds__(256) void k_u64_low_iota(const uint64_t* __restrict__ keys, size_t n, uint32_t* __restrict__ lo, uint32_t* __restrict__ pos) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) lo[i] = (uint32_t)keys[i], pos[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_u64_high_gather(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ pos, size_t n, uint32_t* __restrict__ hi) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) hi[i] = (uint32_t)(keys[pos[i]] >> 32);
}
__global__ __launch_bounds__(256) void k_u64_pairs_gather(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pos, size_t n,
                                                          uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t q = pos[i];
        keys_out[i] = keys[q], vals_out[i] = vals[q];
    }
}
unsigned blocks_for(size_t n) { return (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16); }
}  // namespace

size_t sort_pairs_temp_bytes(size_t n) { return sort_pairs_u32_workspace_bytes(n) + 4 * align_up(n * 4, 256); }

hipError_t sort_pairs_u64_u32(void* temp, size_t temp_bytes, const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                              unsigned end_bit, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (end_bit > 64 || n >= (size_t)kValMask || temp_bytes < sort_pairs_temp_bytes(n)) return hipErrorInvalidValue;
    const size_t ws = sort_pairs_u32_workspace_bytes(n);
    char* w = static_cast<char*>(temp) + ws;
    uint32_t* word = reinterpret_cast<uint32_t*>(w);
    uint32_t* word_s = reinterpret_cast<uint32_t*>(w + align_up(n * 4, 256));
    uint32_t* pos = reinterpret_cast<uint32_t*>(w + 2 * align_up(n * 4, 256));
    uint32_t* pos_s = reinterpret_cast<uint32_t*>(w + 3 * align_up(n * 4, 256));
    hipLaunchKernelGGL(k_u64_low_iota, dim3(blocks_for(n)), dim3(256), 0, stream, keys_in, n, word, pos);
    hipError_t e = sort_pairs_u32_onesweep(temp, ws, word, word_s, pos, pos_s, n, end_bit < 32 ? end_bit : 32u, stream);
    if (e != hipSuccess) return e;
    const uint32_t* order = pos_s;
    if (end_bit > 32) {
        hipLaunchKernelGGL(k_u64_high_gather, dim3(blocks_for(n)), dim3(256), 0, stream, keys_in, pos_s, n, word);
        e = sort_pairs_u32_onesweep(temp, ws, word, word_s, pos_s, pos, n, end_bit - 32u, stream);
        if (e != hipSuccess) return e;
        order = pos;
    }
    hipLaunchKernelGGL(k_u64_pairs_gather, dim3(blocks_for(n)), dim3(256), 0, stream, keys_in, vals_in, order, n, keys_out, vals_out);
    return hipGetLastError();
}

hipError_t sort_pairs_u32_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, size_t n,
                              unsigned end_bit, hipStream_t stream, bool header_zeroed) {
    return sort_pairs_u32_onesweep(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, stream, header_zeroed ? 1 : 0);
}

// ---- prefix scans of int32 arrays: one single-pass kernel, chained through decoupled look-back ------------------------------------------
// A tile of 4096 elements per workgroup, handed out by an atomic ticket; tile t publishes its sum (flag 1) and, once it knows what lies in
// front of it, its inclusive prefix (flag 2); a wave follows the earlier tiles 64 at a time.  State = 1 + tiles words, zeroed per call.
namespace {
constexpr int kScanThreads = 256, kScanItems = 16, kScanTile = kScanThreads * kScanItems;
__global__ __launch_bounds__(kScanThreads) void k_scan_i32(const int32_t* __restrict__ in, int32_t* __restrict__ out, size_t n, int inclusive, unsigned long long* __restrict__ state) {
    __shared__ uint32_t s_tile;
    __shared__ int s_wave[kScanThreads / 64];
    __shared__ int s_excl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_tile = (uint32_t)atomicAdd(state, 1ull);
    __syncthreads();
    const uint32_t tile = s_tile;
    const size_t base = (size_t)tile * kScanTile + (size_t)tid * kScanItems;
    int v[kScanItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) v[k] = base + k < n ? in[base + k] : 0, sum += v[k];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) before += s_wave[w];
        total += s_wave[w];
    }
    if (wave == 0) {
        unsigned long long* st = state + 1;
        int excl = 0;
        if (tile != 0) {
            if (lane == 0) __hip_atomic_store(st + tile, (1ull << 32) | (uint32_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t t = (int64_t)tile - 1;
            while (true) {
                const int64_t mine = t - lane;
                const unsigned long long w = mine >= 0 ? __hip_atomic_load(st + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 32);
                const uint32_t flag = (uint32_t)(w >> 32);
                const unsigned long long ready = __ballot(flag != 0u), prefix = __ballot(flag == 2u);
                const int first_missing = ready == ~0ull ? 64 : __builtin_ctzll(~ready);
                const int first_prefix = prefix == 0ull ? 64 : __builtin_ctzll(prefix);
                const bool done = first_prefix < first_missing;
                const int stop = done ? first_prefix + 1 : first_missing;
                int part = lane < stop ? (int)(uint32_t)w : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                excl += part;
                if (done) break;
                t -= stop;
            }
        }
        if (lane == 0) {
            __hip_atomic_store(st + tile, (2ull << 32) | (uint32_t)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_excl = excl;
        }
    }
    __syncthreads();
    int run = s_excl + before + incl - sum;  // everything in front of this thread's first element
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = inclusive ? run + v[k] : run;
        run += v[k];
    }
}
hipError_t scan_i32(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n, int inclusive, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t tiles = (n + kScanTile - 1) / kScanTile;
    if (temp_bytes < (tiles + 1) * 8) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(temp, 0, (tiles + 1) * 8, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_i32, dim3((unsigned)tiles), dim3(kScanThreads), 0, stream, in, out, n, inclusive, static_cast<unsigned long long*>(temp));
    return hipGetLastError();
}
}  // namespace
size_t scan_temp_bytes(size_t n) { return ((n + kScanTile - 1) / kScanTile + 2) * 8; }
hipError_t inclusive_scan_i32(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t stream) { return scan_i32(temp, temp_bytes, in, out, n, 1, stream); }
hipError_t exclusive_scan_i32(void* temp, size_t temp_bytes, const int32_t* in, int32_t* out, size_t n, hipStream_t stream) { return scan_i32(temp, temp_bytes, in, out, n, 0, stream); }

}  // namespace dmsa
