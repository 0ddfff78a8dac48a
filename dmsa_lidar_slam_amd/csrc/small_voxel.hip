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

// LDS behind the exchange buffer
struct SvShared {
    uint32_t cnt[8][kSvThreads];   // per thread: sixteen 16-bit digit counters (two per word), then its sixteen target offsets; [word][thread]: no bank conflicts
    uint32_t wtot[8][kSvWaves];    // wave totals of the packed counters, then their exclusive prefixes over the waves
    uint32_t tot[8];               // digit totals (packed)
    int scan[4][kSvWaves + 1];     // wave totals of the block-wide scans of the leaf stage
    uint32_t edge[4][kSvWaves + 1];
    int lvl0[2];                   // level 0's totals (Gaussians, members) as level 1 read them
    int fail;
    LatticeTable t;
};

__device__ __forceinline__ uint32_t sv_scan_add_u(uint32_t v) { return (uint32_t)sv_scan_add((int)v); }
// inclusive scan inside each row of 16 lanes
__device__ __forceinline__ uint32_t sv_row_scan_add(uint32_t v) {
    int x = (int)v;
    x += sv_dpp<0x111, 0xf>(x);
    x += sv_dpp<0x112, 0xf>(x);
    x += sv_dpp<0x114, 0xf>(x);
    x += sv_dpp<0x118, 0xf>(x);
    return (uint32_t)x;
}

// K (odd: the blocked reads of the exchange buffer then hit every bank once) positions per thread; thread t owns the positions t K .. t K + K - 1.
template <int K>
__global__ __launch_bounds__(kSvThreads) void k_voxel_small(const SmallVoxelArgs a) {
    static_assert(K % 2 == 1 && K <= 31, "odd, and the flags of a thread's positions are bits of one word");
    constexpr int KH = (K + 1) / 2, KQ = (K + 3) / 4;
    extern __shared__ __align__(16) unsigned char sv_smem[];
    uint32_t* s_buf = reinterpret_cast<uint32_t*>(sv_smem);  // [1024 * K]
    uint16_t* s_buf16 = reinterpret_cast<uint16_t*>(sv_smem);
    SvShared& sh = *reinterpret_cast<SvShared*>(sv_smem + (size_t)kSvThreads * K * 4);
    const int level = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = a.n;
    const int p0 = tid * K;
    int stamp_at = 0;
    auto stamp = [&]() {
        if (a.stamps != nullptr && tid == 0) a.stamps[level * 16 + stamp_at] = (long long)wall_clock64();
        ++stamp_at;
    };
    stamp();  // 0: start
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
            if (level == 0)
                a.counts->pad[0] = 1;
            else
                a.counts->pad[1] = 1;
            if (level == 0) dev_sync_signal(a.sync);
        }
        return;
    }
    const uint32_t invalid = 1u << total_bits;
    const double res = level == 0 ? a.res[0] : a.res[1];

    // ---- leaf codes (k_voxel_keys), point i = r * 1024 + thread: coalesced loads, codes parked in the exchange buffer ----
    {
        const int nev = t.num_events;
        const int nx = t.nbits[0], ny = t.nbits[1], nz = t.nbits[2];
        const int maxb = max(nx, max(ny, nz));
        const int64_t last_ev = nev > 0 ? t.event_idx[nev - 1] : -1;
        // Compressed codes interleave the low nbits of the three keys level by level: one table per axis (key bits -> their places in the code),
        // in the counters' LDS (free until the sort), instead of a loop over the bits per point.
        uint32_t* sp = &sh.cnt[0][0];
        const bool tables = t.compressed && maxb <= 10;
        if (tables) {
            for (int q = tid; q < 3 * 1024; q += kSvThreads) {
                const int ax = q >> 10, v = q & 1023;
                uint32_t out = 0;
                int cur = 0;
                for (int l = 0; l < maxb; ++l) {  // places from the least significant end: z, y, x of level l (the code appends x, y, z from the top)
                    if (l < nz) {
                        if (ax == 2) out |= ((uint32_t)(v >> l) & 1u) << cur;
                        ++cur;
                    }
                    if (l < ny) {
                        if (ax == 1) out |= ((uint32_t)(v >> l) & 1u) << cur;
                        ++cur;
                    }
                    if (l < nx) {
                        if (ax == 0) out |= ((uint32_t)(v >> l) & 1u) << cur;
                        ++cur;
                    }
                }
                sp[q] = out;
            }
        }
        __syncthreads();
        // (double)p - min) / resolution, three divisions per point, is most of this stage on ONE compute unit.  q' = d * (1 / res) differs from
        // the correctly rounded quotient by less than 4e-16 q: wherever q' is further than 1e-9 max(q', 1) from an integer, both truncate to the
        // same key; the division itself only runs for the (one in 10^8) others.
        const double rinv = 1.0 / res;
        auto key_of = [&](float x, double mn, uint32_t mask, uint32_t shiftv) -> uint32_t {
            const double d = (double)x - mn;
            const double q = d * rinv;
            const double f = q - floor(q);
            const double tol = 1e-9 * fmax(q, 1.0);
            uint32_t k;
            if (f > tol && f < 1.0 - tol && q >= 0.0 && q < 4.0e9)
                k = (uint32_t)q;
            else
                k = (uint32_t)(d / res);
            return (k & mask) + shiftv;
        };
        bool oor = false;
#pragma unroll 2
        for (int r = 0; r < K; ++r) {
            const int p = r * kSvThreads + tid;
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
                    const uint32_t kx = key_of(g.x, t.mn[e][0], mask, t.suffix_shift[e][0]);
                    const uint32_t ky = key_of(g.y, t.mn[e][1], mask, t.suffix_shift[e][1]);
                    const uint32_t kz = key_of(g.z, t.mn[e][2], mask, t.suffix_shift[e][2]);
                    if (t.compressed) {
                        if ((kx >> nx) != t.key_base[0] || (ky >> ny) != t.key_base[1] || (kz >> nz) != t.key_base[2]) oor = true;
                        if (tables) {
                            c = sp[kx & ((1u << nx) - 1u)] | sp[1024 + (ky & ((1u << ny) - 1u))] | sp[2048 + (kz & ((1u << nz) - 1u))];
                        } else {
                            uint32_t cc = 0;
                            for (int l = maxb - 1; l >= 0; --l) {
                                if (l < nx) cc = (cc << 1) | ((kx >> l) & 1u);
                                if (l < ny) cc = (cc << 1) | ((ky >> l) & 1u);
                                if (l < nz) cc = (cc << 1) | ((kz >> l) & 1u);
                            }
                            c = cc;
                        }
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
    __syncthreads();
    stamp();  // 1: codes
    // The loops over a thread's K positions are ROLLED (register arrays indexed through the GPR index register, j is uniform): unrolled, the
    // scheduler interleaves the positions and a kernel of 1024 threads (128 registers each) spills by the hundred.  Point indices and
    // positions are below 2^15: two per register; ranks inside a (thread, digit) are below 32: four per register.
    uint32_t key[K], idx2[KH];
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
        key[j] = s_buf[p0 + j];
        const int h16 = (j & 1) * 16;
        idx2[j >> 1] = (idx2[j >> 1] & ~(0xffffu << h16)) | ((uint32_t)(p0 + j) << h16);
    }
    __syncthreads();

    // ---- LSD radix sort of the (code, point) pairs on the code bits [0, end_bit), four bits per pass ----
    // Every thread counts the digits of its own K consecutive positions in sixteen private 16-bit counters (ds_add_rtn returns the rank inside
    // (thread, digit)); the sixteen counter columns are scanned over the threads (two digits per packed word), the digit bases added, and the
    // thread's counters become its target offsets.  Thread-major order inside a digit = position order: stable.
    const int passes = (end_bit + 3) >> 2;
    for (int pass = 0; pass < passes; ++pass) {
        const int shift = 4 * pass;
        const uint32_t dmask = pass == passes - 1 ? ((end_bit - shift) >= 4 ? 15u : ((1u << (end_bit - shift)) - 1u)) : 15u;
#pragma unroll
        for (int w = 0; w < 8; ++w) sh.cnt[w][tid] = 0u;
        uint32_t rk4[KQ], pk[KH];
#pragma unroll 4
        for (int j = 0; j < K; ++j) {
            const uint32_t d = (key[j] >> shift) & dmask;
            const int h16 = (int)(d & 1u) * 16;
            const uint32_t old = atomicAdd(&sh.cnt[d >> 1][tid], 1u << h16);
            const uint32_t r = (old >> h16) & 0xffffu;
            const int b8 = (j & 3) * 8;
            rk4[j >> 2] = (rk4[j >> 2] & ~(0xffu << b8)) | (r << b8);
        }
        uint32_t c[8], incl[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            c[w] = sh.cnt[w][tid];
            incl[w] = sv_scan_add_u(c[w]);
            if (lane == 63) sh.wtot[w][wave] = incl[w];
        }
        __syncthreads();
        if (tid < 8 * kSvWaves) {  // (word, wave) = (tid / 16, tid % 16): exclusive prefix over the waves inside a row of 16 lanes, total of the word
            const int w = tid >> 4, wv = tid & 15;
            const uint32_t v = sh.wtot[w][wv];
            const uint32_t in = sv_row_scan_add(v);
            sh.wtot[w][wv] = in - v;
            if (wv == 15) sh.tot[w] = in;
        }
        __syncthreads();
        {
            uint32_t run = 0;  // base of the digit: total of all smaller digits
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const uint32_t tw = sh.tot[w];
                const uint32_t lo = run, hi = run + (tw & 0xffffu);
                run = hi + (tw >> 16);
                const uint32_t off = (lo | (hi << 16)) + sh.wtot[w][wave] + (incl[w] - c[w]);  // packed: both halves stay below 2^15
                sh.cnt[w][tid] = off;
            }
        }
#pragma unroll 4
        for (int j = 0; j < K; ++j) {
            const uint32_t d = (key[j] >> shift) & dmask;
            const uint32_t off = (sh.cnt[d >> 1][tid] >> ((d & 1u) * 16)) & 0xffffu;
            const uint32_t pos = off + ((rk4[j >> 2] >> ((j & 3) * 8)) & 0xffu);
            s_buf[pos] = key[j];
            const int h16 = (j & 1) * 16;
            pk[j >> 1] = (pk[j >> 1] & ~(0xffffu << h16)) | (pos << h16);
        }
        __syncthreads();
        if (pass == 0) stamp();  // 2: first pass up to the scatter of the codes
#pragma unroll 4
        for (int j = 0; j < K; ++j) key[j] = s_buf[p0 + j];
        if (pass == passes - 1)
            for (int i = tid; i < n; i += kSvThreads) g_code_s[i] = s_buf[i];
        __syncthreads();
#pragma unroll 4
        for (int j = 0; j < K; ++j) {
            const int h16 = (j & 1) * 16;
            s_buf16[(pk[j >> 1] >> h16) & 0xffffu] = (uint16_t)(idx2[j >> 1] >> h16);
        }
        __syncthreads();
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            const int h16 = (j & 1) * 16;
            idx2[j >> 1] = (idx2[j >> 1] & ~(0xffffu << h16)) | ((uint32_t)s_buf16[p0 + j] << h16);
        }
        if (pass == passes - 1)
            for (int i = tid; i < n; i += kSvThreads) g_idx_s[i] = (uint32_t)s_buf16[i];
        if (pass == 0) stamp();  // 3: first pass done
    }
    __syncthreads();
    stamp();  // 4: sorted
    auto idx_of = [&](int j) -> uint32_t { return (idx2[j >> 1] >> ((j & 1) * 16)) & 0xffffu; };

    // ---- leaves.  A thread walks its K consecutive positions; what it needs from the threads in front comes from block-wide scans. ----
    uint32_t validm = 0, headm = 0, diffm = 0, tailm = 0;  // bit j: flag of position p0 + j
    {
        const uint32_t lastk = key[K - 1];
        const bool lastv = lastk < invalid;
        const int lastid = lastv ? a.ring[idx_of(K - 1)] : 0;
        uint32_t pkey = (uint32_t)__shfl_up((int)lastk, 1);
        int pid = __shfl_up(lastid, 1);
        if (lane == 63) sh.edge[0][wave + 1] = lastk, sh.edge[1][wave + 1] = (uint32_t)lastid;
        __syncthreads();
        if (lane == 0 && wave > 0) pkey = sh.edge[0][wave], pid = (int)sh.edge[1][wave];
        // (thread 0's position 0 is a head whatever pkey holds)
#pragma unroll 1
        for (int j0 = 0; j0 < K; j0 += 4) {
            int idc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                idc[u] = (j < K && key[j < K ? j : 0] < invalid) ? a.ring[idx_of(j < K ? j : 0)] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                if (j < K) {
                    const uint32_t kj = key[j];
                    const bool v = kj < invalid;
                    const bool h = v && ((p0 + j) == 0 || pkey != kj);
                    validm |= v ? (1u << j) : 0u;
                    headm |= h ? (1u << j) : 0u;
                    diffm |= (v && !h && pid != idc[u]) ? (1u << j) : 0u;
                    pkey = kj, pid = idc[u];
                }
            }
        }
    }
    stamp();  // 5: head flags, ring ids
    // carry-in 1: start position of the leaf that holds the position in front of the thread's first (exclusive max-scan of the threads' last heads)
    int carry_start;
    {
        const int last_head = headm != 0u ? p0 + (31 - __builtin_clz(headm)) : 0;
        const int in = sv_scan_max(last_head);
        if (lane == 63) sh.scan[0][wave + 1] = in;
        const int heads_w = sv_scan_add(__popc(headm));
        if (lane == 63) sh.scan[1][wave] = heads_w;
        // the position behind the thread's last: a head, or not a valid position?
        if (lane == 0) sh.edge[2][wave] = headm & 1u, sh.edge[3][wave] = validm & 1u;
        if (tid == 0) sh.edge[2][kSvWaves] = 1u, sh.edge[3][kSvWaves] = 0u, sh.scan[0][0] = 0;
        __syncthreads();
        int before = 0;
        for (int w = 0; w <= wave; ++w) before = max(before, sh.scan[0][w]);
        int ex = __shfl_up(in, 1);
        if (lane == 0) ex = 0;
        carry_start = max(before, ex);
    }
    int num_leaves = 0;
    for (int w = 0; w < kSvWaves; ++w) num_leaves += sh.scan[1][w];
    // tails: the position behind is a head or no valid position at all
    {
        uint32_t nh = (uint32_t)__shfl_down((int)(headm & 1u), 1), nv = (uint32_t)__shfl_down((int)(validm & 1u), 1);
        if (lane == 63) nh = sh.edge[2][wave + 1], nv = sh.edge[3][wave + 1];
        const uint32_t next_head = (headm >> 1) | (nh << (K - 1)), next_valid = (validm >> 1) | (nv << (K - 1));
        tailm = validm & (next_head | ~next_valid) & ((K == 32) ? 0xffffffffu : ((1u << K) - 1u));
    }
    // carry-in 2: have the ids of that leaf differed so far?  Every thread offers (start of its last position's leaf << 1 | ids differed in the part
    // of that leaf it holds); the exclusive max-scan ORs the bits of the threads that share the leaf.
    int carry_mixed;
    {
        int cur = carry_start;
        uint32_t dif = 0;
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            if ((headm >> j) & 1u) cur = p0 + j, dif = 0;
            dif |= (diffm >> j) & 1u;
        }
        const int val = (validm & 1u) ? ((cur << 1) | (int)dif) : 0;  // (threads behind the last valid position offer nothing)
        const int in = sv_scan_max(val);
        if (lane == 63) sh.scan[2][wave + 1] = in;
        if (tid == 0) sh.scan[2][0] = 0;
        __syncthreads();
        int before = 0;
        for (int w = 0; w <= wave; ++w) before = max(before, sh.scan[2][w]);
        int ex = __shfl_up(in, 1);
        if (lane == 0) ex = 0;
        const int c = max(before, ex);
        carry_mixed = ((c >> 1) == carry_start) ? (c & 1) : 0;
    }
    stamp();  // 6: carries
    // acceptance at the leaf's last position (DmsaOptimizer.h:302-307): the thread's accepted leaves and their members, packed
    uint32_t accm = 0;
    int mine = 0;
    {
        int cur = carry_start;
        int dif = carry_mixed;
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            if ((headm >> j) & 1u) cur = p0 + j, dif = 0;
            dif |= (int)((diffm >> j) & 1u);
            if ((tailm >> j) & 1u) {
                const int size = p0 + j - cur + 1;
                if (size >= a.min_pts && dif != 0) accm |= 1u << j, mine += 1 | (size << 15);  // accepted leaves in bits 0 .. 14 (<= n / 2), members above
            }
        }
    }
    int before_acc, num_gauss, num_memb;
    {
        const int in = sv_scan_add(mine);
        if (lane == 63) sh.scan[3][wave] = in;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < kSvWaves; ++w) {
            if (w < wave) before += sh.scan[3][w];
            total += sh.scan[3][w];
        }
        before_acc = before + in - mine;
        num_gauss = total & 0x7fff, num_memb = total >> 15;
    }
    stamp();  // 7: acceptance
    int gbase = 0, mbase = 0;
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
    stamp();  // 8: level 0 published / level 1 has waited
    // Gaussian index and member offset of every accepted leaf, left at the leaf's start position in the exchange buffer for the threads that hold
    // its other positions: accepted | Gaussian (15 bits) | member offset (16 bits)
    {
        int cur = carry_start, run = before_acc;
#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            if ((headm >> j) & 1u) cur = p0 + j;
            if ((tailm >> j) & 1u) {
                const bool acc = (accm >> j) & 1u;
                s_buf[cur] = acc ? (0x80000000u | ((uint32_t)(run & 0x7fff) << 16) | (uint32_t)(run >> 15)) : 0u;
                if (acc) run += 1 | ((p0 + j - cur + 1) << 15);
            }
        }
    }
    __syncthreads();
    if (level == 1) {
        if (sh.fail != 0) return;  // level 0 gave up: the host repeats the voxelisation on the general path
        gbase = sh.lvl0[0], mbase = sh.lvl0[1];
    }
    stamp();  // 9: offsets
    // ---- member lists (k_gather_members) ----
    {
        int cur = carry_start;
        uint32_t info = (validm & 1u) ? s_buf[carry_start] : 0u;
#pragma unroll 2
        for (int j = 0; j < K; ++j) {
            if ((headm >> j) & 1u) cur = p0 + j, info = s_buf[cur];
            if (((validm >> j) & 1u) && (info >> 31)) {
                const uint32_t pi = idx_of(j);
                const float4 loc = a.local[pi];
                const int g = gbase + (int)((info >> 16) & 0x7fffu);
                const int rank = p0 + j - cur;
                const int dst = mbase + (int)(info & 0xffffu) + rank;
                a.memb_local[dst] = loc;
                a.memb_idx[dst] = (int32_t)pi;
                a.memb_g[dst] = (int32_t)((uint32_t)g | (((tailm >> j) & 1u) ? 0x80000000u : 0u));
                if (rank == 0) a.seg_off[g] = dst;
            }
        }
    }
    stamp();  // 10: member lists
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

int small_voxel_max_points() { return kSvThreads * 29; }  // 29 positions per thread: 116 KB of exchange buffer + 37 KB of counters and tables in 160 KB of LDS

void launch_voxel_small(const SmallVoxelArgs& a, hipStream_t s) {
    const int k = (a.n + kSvThreads - 1) / kSvThreads;
    if (k <= 9)
        launch_k<9>(a, s);
    else if (k <= 17)
        launch_k<17>(a, s);
    else if (k <= 25)
        launch_k<25>(a, s);
    else
        launch_k<29>(a, s);
}

}  // namespace dmsa
