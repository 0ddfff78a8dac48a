// small_voxel.hip — createGaussianSets at BOTH resolutions (DmsaOptimizer.h:275-350 + PCL OctreePointCloud) for the reference's everyday
// problem size in ONE launch: 5 scans x <= 3000 points + <= 10^4 static points (config/slam_settings.yaml:6, config/livox.yaml:43).
//
// The general path (voxelize_driver.cpp) is ten dependent kernels per level, tuned for 1.5 x 10^6 points: at 25 000 points every one of them is
// a latency floor of 5-15 us (launch, memory round trips, end-of-kernel write-back) and the stage costs ~90 us.  Here ONE workgroup of 1024
// threads per level keeps its level's (code, point) pairs in REGISTERS -- thread (wave w, lane l) owns the positions w * 64 K + j * 64 + l,
// j < K -- and does everything between the lattice and the member lists without leaving the compute unit:
//   leaf codes (genOctreeKeyforPoint with the box in force when the point was inserted; same expressions as k_voxel_keys)
//   -> LSD radix sort, 8 bits per pass, only as many passes as the tree's code width needs: per wave the ranks of equal digits by
//      match-any ballots over the lanes + running digit counters in LDS (stable: position order), a column scan over the 16 waves, the
//      pairs exchanged through one LDS buffer (codes, then point indices: 4 bytes x n is all that fits beside the counters)
//   -> head flags, start position of every position's leaf (max-scan), "ids differ inside the leaf" (adjacent comparison, max-scan),
//      acceptance `size >= min && max(id) != min(id)` (:307) at the leaf's last position, Gaussian / member offsets (sum-scan)
//   -> member lists in Gaussian order (leaf depth-first order, ascending point index inside a leaf), seg_off, counts.
// Level 1's Gaussians follow level 0's: its workgroup takes level 0's totals from device memory behind a counter (dev_sync.h), the one
// cross-workgroup dependency.  Same bits as the general path (tests/test_gpu_small_voxel.py compares both on the same contexts).
#include "dmsa_kernels.h"

#include "dev_sync.h"

#include <hip/hip_runtime.h>

#include <cstdint>

namespace dmsa {
namespace {

template <int kCtrl, int kRowMask>
__device__ __forceinline__ int sv_dpp(int v) {
    return __builtin_amdgcn_update_dpp(0, v, kCtrl, kRowMask, 0xf, true);
}
__device__ __forceinline__ int sv_scan_add(int v) {  // inclusive over the wave's lanes
    v += sv_dpp<0x111, 0xf>(v);
    v += sv_dpp<0x112, 0xf>(v);
    v += sv_dpp<0x114, 0xf>(v);
    v += sv_dpp<0x118, 0xf>(v);
    v += sv_dpp<0x142, 0xa>(v);
    v += sv_dpp<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ int sv_scan_max(int v) {  // inclusive, values >= 0 (lanes that receive nothing read 0)
    v = max(v, sv_dpp<0x111, 0xf>(v));
    v = max(v, sv_dpp<0x112, 0xf>(v));
    v = max(v, sv_dpp<0x114, 0xf>(v));
    v = max(v, sv_dpp<0x118, 0xf>(v));
    v = max(v, sv_dpp<0x142, 0xa>(v));
    v = max(v, sv_dpp<0x143, 0xc>(v));
    return v;
}
__device__ __forceinline__ uint64_t sv_spread3(uint32_t v) {  // 21 bits -> every third bit
    uint64_t x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

constexpr int kSvThreads = 1024, kSvWaves = 16;

// LDS behind the exchange buffer (uint32 words)
struct SvShared {
    uint32_t cnt[kSvWaves][256];  // per wave: running digit counters, then the wave's offset inside its digit
    uint32_t base[256];           // first position of every digit
    int wtot[6][kSvWaves];        // wave totals of the block-wide scans
    uint32_t edge[4][kSvWaves + 1];
    int lvl0[2];                  // level 0's totals (Gaussians, members) as level 1 read them
    int fail;
    LatticeTable t;
};

template <int K>
__global__ __launch_bounds__(kSvThreads) void k_voxel_small(const SmallVoxelArgs a) {
    extern __shared__ __align__(16) unsigned char sv_smem[];
    uint32_t* s_buf = reinterpret_cast<uint32_t*>(sv_smem);  // [1024 * K]
    SvShared& sh = *reinterpret_cast<SvShared*>(sv_smem + (size_t)kSvThreads * K * 4);
    const int level = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = a.n;
    const int wbase = wave * 64 * K;
    LatticeTable* table = a.tables + level;
    // (selected, not indexed: a dynamic index into the by-value argument struct would copy it to scratch memory)
    uint32_t* const g_code = level == 0 ? a.code[0] : a.code[1];
    uint32_t* const g_idx = level == 0 ? a.idx[0] : a.idx[1];
    uint32_t* const g_code_s = level == 0 ? a.code_s[0] : a.code_s[1];
    uint32_t* const g_idx_s = level == 0 ? a.idx_s[0] : a.idx_s[1];
    {
        const int words = sizeof(LatticeTable) / 4;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(table);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&sh.t);
        for (int i = tid; i < words; i += kSvThreads) dst[i] = src[i];
        for (int q = tid; q < kSvWaves * 256; q += kSvThreads) (&sh.cnt[0][0])[q] = 0u;
        if (tid == 0) sh.fail = 0;
    }
    __syncthreads();
    const LatticeTable& t = sh.t;
    if (tid == 0) {
        sh.t.code_or = 0ull;
        table->code_or = 0ull;  // (dmsa_get_voxel_level strips the tag of the merged sort: there is none here)
    }
    const int total_bits = t.compressed ? t.total_bits : 3 * t.final_depth;
    const int end_bit = total_bits + 1;  // bit `total_bits` marks the non-finite points
    if (t.status != 0 || end_bit > 32 || total_bits < 0) {
        // codes wider than 32 bits (or a tree deeper than PCL allows): the host runs the general path again
        if (tid == 0) {
            if (level == 0) a.counts->pad[0] = 1; else a.counts->pad[1] = 1;
            if (level == 0) dev_sync_signal(a.sync);
        }
        return;
    }
    __syncthreads();
    const uint32_t invalid = total_bits >= 32 ? 0u : (1u << total_bits);
    const double res = level == 0 ? a.res[0] : a.res[1];

    // ---- leaf codes (k_voxel_keys) ----
    // (computed in a rolled loop -- three fp64 divisions per point, and fully unrolled the compiler keeps every point of the lane in flight --
    // and parked in the exchange buffer; every lane reads back what it wrote itself)
    {
        const int nev = t.num_events;
        const int nx = t.nbits[0], ny = t.nbits[1], nz = t.nbits[2];
        const int maxb = max(nx, max(ny, nz));
        const int64_t last_ev = nev > 0 ? t.event_idx[nev - 1] : -1;
        bool oor = false;
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const int p = wbase + j * 64 + lane;
            uint32_t c = 0xffffffffu;  // padding behind the last point: sorts behind everything
            if (p < n) {
                const float4 g = a.global[p];
                c = invalid;
                if (isfinite(g.x) && isfinite(g.y) && isfinite(g.z)) {
                    int e = nev;
                    if (p < last_ev) {
                        e = 0;
                        while (e < nev && t.event_idx[e] <= p) ++e;
                    }
                    const uint32_t mask = (1u << t.depth[e]) - 1u;
                    const uint32_t kx = ((uint32_t)(((double)g.x - t.mn[e][0]) / res) & mask) + t.suffix_shift[e][0];
                    const uint32_t ky = ((uint32_t)(((double)g.y - t.mn[e][1]) / res) & mask) + t.suffix_shift[e][1];
                    const uint32_t kz = ((uint32_t)(((double)g.z - t.mn[e][2]) / res) & mask) + t.suffix_shift[e][2];
                    if (t.compressed) {
                        if ((kx >> nx) != t.key_base[0] || (ky >> ny) != t.key_base[1] || (kz >> nz) != t.key_base[2]) oor = true;
                        uint32_t cc = 0;
                        for (int l = maxb - 1; l >= 0; --l) {
                            if (l < nx) cc = (cc << 1) | ((kx >> l) & 1u);
                            if (l < ny) cc = (cc << 1) | ((ky >> l) & 1u);
                            if (l < nz) cc = (cc << 1) | ((kz >> l) & 1u);
                        }
                        c = cc;
                    } else {
                        c = (uint32_t)((sv_spread3(kx) << 2) | (sv_spread3(ky) << 1) | sv_spread3(kz));
                    }
                }
                g_code[p] = c;
                g_idx[p] = (uint32_t)p;
            }
            s_buf[p] = c;
        }
        if (oor) table->out_of_range = 1;
    }
    // From here on the loops over a lane's K positions are ROLLED (the register arrays are indexed through the GPR index register, j is
    // uniform): unrolled, the scheduler interleaves the positions and a kernel of 1024 threads (128 registers each) spills by the hundred.
    // Point indices, target positions and start positions are all below 2^15: two per register.
    uint32_t key[K], idx2[K / 2];
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
        const int p = wbase + j * 64 + lane;
        key[j] = s_buf[p];
        const int h16 = (j & 1) * 16;
        idx2[j >> 1] = (idx2[j >> 1] & ~(0xffffu << h16)) | ((uint32_t)p << h16);
    }
    __syncthreads();
    uint16_t* s_buf16 = reinterpret_cast<uint16_t*>(s_buf);

    // ---- LSD radix sort of the (code, point) pairs on the code bits [0, end_bit) ----
    const int passes = (end_bit + 7) >> 3;
    const uint64_t lanes_below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = 8 * pass;
        const uint32_t dmask = pass == passes - 1 ? ((end_bit - shift) >= 8 ? 255u : ((1u << (end_bit - shift)) - 1u)) : 255u;
        uint32_t* cnt = sh.cnt[wave];
        uint32_t pk[K / 2];  // rank inside (wave, digit), then the target position
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const uint32_t d = (key[j] >> shift) & dmask;
            uint64_t m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const int rk = __popcll(m & lanes_below);
            const int leader = __ffsll((long long)m) - 1;
            uint32_t first = 0;
            if (lane == leader) {
                first = cnt[d];
                cnt[d] = first + (uint32_t)__popcll(m);
            }
            first = (uint32_t)__shfl((int)first, leader);
            const uint32_t r = first + (uint32_t)rk;
            const int h16 = (j & 1) * 16;
            pk[j >> 1] = (pk[j >> 1] & ~(0xffffu << h16)) | (r << h16);
        }
        __syncthreads();
        // offsets: digit-major, wave-minor
        int tot = 0, incl = 0;
        if (tid < 256) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < kSvWaves; ++w) {
                const uint32_t c = sh.cnt[w][tid];
                sh.cnt[w][tid] = run;
                run += c;
            }
            tot = (int)run;
            incl = sv_scan_add(tot);
            if (lane == 63) sh.wtot[0][wave] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            int before = 0;
            for (int w = 0; w < wave; ++w) before += sh.wtot[0][w];
            sh.base[tid] = (uint32_t)(before + incl - tot);
        }
        __syncthreads();
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const uint32_t d = (key[j] >> shift) & dmask;
            const int h16 = (j & 1) * 16;
            const uint32_t r = (pk[j >> 1] >> h16) & 0xffffu;
            const uint32_t pos = sh.base[d] + cnt[d] + r;
            s_buf[pos] = key[j];
            pk[j >> 1] = (pk[j >> 1] & ~(0xffffu << h16)) | (pos << h16);
        }
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < K; ++j) key[j] = s_buf[wbase + j * 64 + lane];
        for (int q = tid; q < kSvWaves * 256; q += kSvThreads) (&sh.cnt[0][0])[q] = 0u;  // for the next pass
        __syncthreads();
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const int h16 = (j & 1) * 16;
            s_buf16[(pk[j >> 1] >> h16) & 0xffffu] = (uint16_t)(idx2[j >> 1] >> h16);
        }
        __syncthreads();
#pragma unroll 2
        for (int j = 0; j < K; j += 2)
            idx2[j >> 1] = (uint32_t)s_buf16[wbase + j * 64 + lane] | ((uint32_t)s_buf16[wbase + (j + 1) * 64 + lane] << 16);
        __syncthreads();
    }
    auto idx_of = [&](int j) -> uint32_t { return (idx2[j >> 1] >> ((j & 1) * 16)) & 0xffffu; };
#pragma unroll 2
    for (int j = 0; j < K; ++j) {
        const int p = wbase + j * 64 + lane;
        if (p < n) g_code_s[p] = key[j], g_idx_s[p] = idx_of(j);
    }

    // ---- leaves: head flags from the codes ----
    uint32_t validm = 0, headm = 0, diffm = 0, tailm = 0;  // bit j: flag of position (j, lane)
    if (lane == 63) sh.edge[0][wave + 1] = key[K - 1];
    {   // the last position's ring id goes to the next wave as well
        const bool v = key[K - 1] < invalid;
        const int idl = v ? a.ring[idx_of(K - 1)] : 0;
        if (lane == 63) sh.edge[1][wave + 1] = (uint32_t)idl;
    }
    __syncthreads();
    int heads = 0;
    {
        uint32_t carry = wave > 0 ? sh.edge[0][wave] : 0u;  // code of the position in front of (j, lane 0)
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const int p = wbase + j * 64 + lane;
            const uint32_t kj = key[j];
            uint32_t pkey = (uint32_t)__shfl_up((int)kj, 1);
            if (lane == 0) pkey = carry;
            carry = (uint32_t)__builtin_amdgcn_readlane((int)kj, 63);
            const bool v = kj < invalid;
            const bool h = v && (p == 0 || pkey != kj);
            validm |= v ? (1u << j) : 0u;
            headm |= h ? (1u << j) : 0u;
            heads += __popcll(__ballot(h));
        }
    }
    // ---- "differs from the position before" from the ring ids, four positions of the lane in flight ----
    {
        int carry = wave > 0 ? (int)sh.edge[1][wave] : 0;  // id of the position in front of (j, lane 0)
#pragma unroll 1
        for (int j0 = 0; j0 < K; j0 += 4) {
            int idc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) idc[u] = ((validm >> (j0 + u)) & 1u) ? a.ring[idx_of(j0 + u)] : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                int pid = __shfl_up(idc[u], 1);
                if (lane == 0) pid = carry;
                carry = __builtin_amdgcn_readlane(idc[u], 63);
                const bool v = (validm >> j) & 1u, h = (headm >> j) & 1u;
                diffm |= (v && !h && pid != idc[u]) ? (1u << j) : 0u;
            }
        }
    }
    // ---- start position of the leaf of every position: inclusive max-scan of (head ? position : 0); kept 16 bits wide ----
    uint32_t st2[K / 2];
    {
        int run = 0;
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const int p = wbase + j * 64 + lane;
            const int sc = sv_scan_max(((headm >> j) & 1u) ? p : 0);
            const uint32_t stj = (uint32_t)max(run, sc);
            run = max(run, __builtin_amdgcn_readlane(sc, 63));
            const int h16 = (j & 1) * 16;
            st2[j >> 1] = (st2[j >> 1] & ~(0xffffu << h16)) | (stj << h16);
        }
        if (lane == 0) sh.wtot[1][wave] = run, sh.wtot[2][wave] = heads;
        if (lane == 0) sh.edge[2][wave] = headm & 1u, sh.edge[3][wave] = validm & 1u;  // first position of the wave: head? valid?
        if (tid == 0) sh.edge[2][kSvWaves] = 1u, sh.edge[3][kSvWaves] = 0u;           // behind the last position: nothing
    }
    __syncthreads();
    int num_leaves = 0, before_start = 0;
    for (int w = 0; w < kSvWaves; ++w) {
        if (w < wave) before_start = max(before_start, sh.wtot[1][w]);
        num_leaves += sh.wtot[2][w];
    }
    // (a leaf that began in an earlier wave: the wave's own scan says 0 until its first head)
    auto start_of = [&](int j) -> int { return max((int)((st2[j >> 1] >> ((j & 1) * 16)) & 0xffffu), before_start); };

    // "ids differ inside the leaf so far": inclusive max-scan of (start << 1 | differs from the position before).  Only the low bit is kept
    // (mixm): the part of the leaf that lies in earlier waves is added from the waves' carries below.  Tails: the position behind is a head or
    // no valid position at all.
    uint32_t mixm = 0;
    {
        int run = 0;
        const uint32_t head0 = (uint32_t)__builtin_amdgcn_readlane((int)headm, 0), valid0 = (uint32_t)__builtin_amdgcn_readlane((int)validm, 0);
        const uint32_t next_h = sh.edge[2][wave + 1], next_v = sh.edge[3][wave + 1];
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const bool v = (validm >> j) & 1u;
            const int sc = sv_scan_max(v ? ((start_of(j) << 1) | (int)((diffm >> j) & 1u)) : 0);
            const int mx = max(run, sc);  // (mx >> 1) == start for every valid position: its own entry takes part
            mixm |= (mx & 1) ? (1u << j) : 0u;
            run = max(run, __builtin_amdgcn_readlane(sc, 63));
            uint32_t nh = (uint32_t)__shfl_down((int)((headm >> j) & 1u), 1), nv = (uint32_t)__shfl_down((int)((validm >> j) & 1u), 1);
            if (lane == 63) {
                nh = j + 1 < K ? ((head0 >> (j + 1)) & 1u) : next_h;
                nv = j + 1 < K ? ((valid0 >> (j + 1)) & 1u) : next_v;
            }
            tailm |= (v && (nh != 0u || nv == 0u)) ? (1u << j) : 0u;
        }
        if (lane == 0) sh.wtot[3][wave] = run;
    }
    __syncthreads();
    // acceptance at the leaf's last position (DmsaOptimizer.h:302-307)
    uint32_t accm = 0;
    {
        int before = 0;
        for (int w = 0; w < wave; ++w) before = max(before, sh.wtot[3][w]);
        int run = 0;
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const int p = wbase + j * 64 + lane;
            const bool tl = (tailm >> j) & 1u;
            const int stj = start_of(j);
            const int size = p - stj + 1;
            const bool mixed = ((mixm >> j) & 1u) != 0u || ((before >> 1) == stj && (before & 1) != 0);
            const bool acc = tl && size >= a.min_pts && mixed;
            accm |= acc ? (1u << j) : 0u;
            const int v = acc ? (1 | (size << 15)) : 0;  // accepted leaves in bits 0 .. 14 (<= n / 2), their members above
            run += __builtin_amdgcn_readlane(sv_scan_add(v), 63);
        }
        if (lane == 0) sh.wtot[4][wave] = run;
    }
    __syncthreads();
    int gbase = 0, mbase = 0, num_gauss, num_memb, before_acc = 0;
    {
        int total = 0;
        for (int w = 0; w < kSvWaves; ++w) {
            if (w < wave) before_acc += sh.wtot[4][w];
            total += sh.wtot[4][w];
        }
        num_gauss = total & 0x7fff, num_memb = total >> 15;
    }
    if (level == 0) {
        if (tid == 0) {
            a.counts->level[0].num_leaves = num_leaves, a.counts->level[0].num_gauss = num_gauss, a.counts->level[0].num_memb = num_memb;
            a.counts->level[0].pad = 0;
            dev_sync_signal(a.sync);
        }
    } else {
        if (tid == 0) {
            dev_sync_wait(a.sync, a.sync_target, a.timed_out);
            sh.lvl0[0] = __hip_atomic_load(&a.counts->level[0].num_gauss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.lvl0[1] = __hip_atomic_load(&a.counts->level[0].num_memb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh.fail = __hip_atomic_load(&a.counts->pad[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // Gaussian index and member offset of every accepted leaf (exclusive sum-scan over the positions), left at the leaf's start position in
    // the exchange buffer for all its positions: accepted | Gaussian (15 bits) | member offset (16 bits)
    {
        int run = before_acc;
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            const int p = wbase + j * 64 + lane;
            const bool acc = (accm >> j) & 1u;
            const int stj = start_of(j);
            const int v = acc ? (1 | ((p - stj + 1) << 15)) : 0;
            const int sc = sv_scan_add(v);
            const int ex = run + sc - v;
            run += __builtin_amdgcn_readlane(sc, 63);
            if ((tailm >> j) & 1u) s_buf[stj] = acc ? (0x80000000u | ((uint32_t)(ex & 0x7fff) << 16) | (uint32_t)(ex >> 15)) : 0u;
        }
    }
    __syncthreads();
    if (level == 1) {
        if (sh.fail != 0) return;  // level 0 gave up: the host repeats the voxelisation on the general path
        gbase = sh.lvl0[0], mbase = sh.lvl0[1];
    }
    // ---- member lists (k_gather_members), four positions per lane in flight (named variables: a small array indexed in an inner loop
    // ends up in scratch memory here) ----
    auto member_info = [&](int j) -> uint32_t { return ((validm >> j) & 1u) ? s_buf[start_of(j)] : 0u; };
    auto member_load = [&](int j, uint32_t info) -> float4 { return (info >> 31) ? a.local[idx_of(j)] : make_float4(0.f, 0.f, 0.f, 0.f); };
    auto member_store = [&](int j, uint32_t info, const float4 loc) {
        if (info >> 31) {
            const int p = wbase + j * 64 + lane;
            const int g = gbase + (int)((info >> 16) & 0x7fffu);
            const int rank = p - start_of(j);
            const int dst = mbase + (int)(info & 0xffffu) + rank;
            a.memb_local[dst] = loc;
            a.memb_idx[dst] = (int32_t)idx_of(j);
            a.memb_g[dst] = (int32_t)((uint32_t)g | (((tailm >> j) & 1u) ? 0x80000000u : 0u));
            if (rank == 0) a.seg_off[g] = dst;
        }
    };
#pragma unroll 1
    for (int j0 = 0; j0 < K; j0 += 4) {
        const uint32_t i0 = member_info(j0), i1 = member_info(j0 + 1), i2 = member_info(j0 + 2), i3 = member_info(j0 + 3);
        const float4 l0 = member_load(j0, i0), l1 = member_load(j0 + 1, i1), l2 = member_load(j0 + 2, i2), l3 = member_load(j0 + 3, i3);
        member_store(j0, i0, l0), member_store(j0 + 1, i1, l1), member_store(j0 + 2, i2, l2), member_store(j0 + 3, i3, l3);
    }
    if (tid == 0) {
        a.seg_off[gbase + num_gauss] = mbase + num_memb;
        if (level == 1) {
            a.counts->level[1].num_leaves = num_leaves, a.counts->level[1].num_gauss = num_gauss, a.counts->level[1].num_memb = num_memb;
            a.counts->level[1].pad = 0;
        }
    }
}

template <int K>
void launch_k(const SmallVoxelArgs& a, hipStream_t s) {
    const size_t smem = (size_t)kSvThreads * K * 4 + sizeof(SvShared);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_voxel_small<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_voxel_small<K>, dim3(2), dim3(kSvThreads), smem, s, a);
}

}  // namespace

int small_voxel_max_points() { return kSvThreads * 32; }

void launch_voxel_small(const SmallVoxelArgs& a, hipStream_t s) {
    const int k = (a.n + kSvThreads - 1) / kSvThreads;
    if (k <= 8)
        launch_k<8>(a, s);
    else if (k <= 16)
        launch_k<16>(a, s);
    else if (k <= 24)
        launch_k<24>(a, s);
    else if (k <= 28)
        launch_k<28>(a, s);
    else
        launch_k<32>(a, s);
}

}  // namespace dmsa
