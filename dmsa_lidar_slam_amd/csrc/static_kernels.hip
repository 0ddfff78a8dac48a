// static_kernels.hip — see static_kernels.h.  Compiled with -ffp-contract=off: the float distance
// ((dx*dx + dy*dy) + dz*dz) and the visibility residual must round like the reference's scalar code.
#include "static_kernels.h"

#include "../../include/dmsa_detmath.h"

#include <cfloat>
#include <climits>

namespace dmsa {
namespace {
constexpr int kBlock = 256;
inline unsigned grid_for(int64_t n, int block = kBlock, int64_t cap = 1 << 20) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}
__device__ __forceinline__ bool finite3(const float4 p) { return isfinite(p.x) && isfinite(p.y) && isfinite(p.z); }
__device__ __forceinline__ uint64_t hash_cell(uint64_t key) {
    key *= 0x9E3779B97F4A7C15ull;
    return key ^ (key >> 29);
}
__device__ __forceinline__ bool cell_of(const float4 p, const CellGrid& g, int64_t& ix, int64_t& iy, int64_t& iz) {
    if (!finite3(p)) return false;
    const double vx = ((double)p.x - g.lo[0]) * g.inv, vy = ((double)p.y - g.lo[1]) * g.inv, vz = ((double)p.z - g.lo[2]) * g.inv;
    // more than one cell outside the grid: no cell of the cloud can be a neighbour (also keeps the casts in range)
    if (!(vx > -2.0 && vy > -2.0 && vz > -2.0 && vx < (double)g.nx + 2.0 && vy < (double)g.ny + 2.0 && vz < (double)g.nz + 2.0)) return false;
    ix = (int64_t)floor(vx), iy = (int64_t)floor(vy), iz = (int64_t)floor(vz);
    return true;
}
}  // namespace

__global__ void k_cloud_bounds_init(CloudBounds* b) {
    if (threadIdx.x < 3) b->lo[threadIdx.x] = 0xFFFFFFFFu, b->hi[threadIdx.x] = 0u;
    if (threadIdx.x == 0) b->num_finite = 0u, b->pad = 0u;
}
__global__ __launch_bounds__(kBlock) void k_cloud_bounds(const float4* __restrict__ pts, int64_t n, CloudBounds* __restrict__ b) {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u}, cnt = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 p = pts[i];
        if (!finite3(p)) continue;
        const uint32_t o[3] = {float_to_ordered(p.x), float_to_ordered(p.y), float_to_ordered(p.z)};
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = min(lo[a], o[a]), hi[a] = max(hi[a], o[a]);
        ++cnt;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], m)), hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], m));
        cnt += (uint32_t)__shfl_xor((int)cnt, m);
    }
    // one set of atomics per WORKGROUP: all of them land on one cache line, and 57 k same-line atomics (one set per wave of a
    // 2048-block grid) cost 0.6 ms on a 1.3 M-point cloud
    __shared__ uint32_t s_part[kBlock / 64][7];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) s_part[wave][a] = lo[a], s_part[wave][3 + a] = hi[a];
        s_part[wave][6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) {
#pragma unroll
            for (int a = 0; a < 3; ++a) lo[a] = min(lo[a], s_part[w][a]), hi[a] = max(hi[a], s_part[w][3 + a]);
            cnt += s_part[w][6];
        }
        if (cnt > 0) {
#pragma unroll
            for (int a = 0; a < 3; ++a) atomicMin(&b->lo[a], lo[a]), atomicMax(&b->hi[a], hi[a]);
            atomicAdd(&b->num_finite, cnt);
        }
    }
}
void launch_cloud_bounds_init(CloudBounds* b, hipStream_t s) { hipLaunchKernelGGL(k_cloud_bounds_init, dim3(1), dim3(64), 0, s, b); }
void launch_cloud_bounds(const float4* pts, int64_t n, CloudBounds* b, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_cloud_bounds, dim3(grid_for(n, kBlock, 512)), dim3(kBlock), 0, s, pts, n, b);
}

template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_cell_codes(const float4* __restrict__ pts, int64_t n, CellGrid g, KeyT* __restrict__ code,
                                                       uint32_t* __restrict__ idx) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        int64_t ix, iy, iz;
        KeyT c = (KeyT)~(KeyT)0;
        if (cell_of(pts[i], g, ix, iy, iz)) {
            // the grid was sized from the bounds of the same points: clamp only guards the last ulp of the division
            ix = min(max(ix, (int64_t)0), g.nx - 1), iy = min(max(iy, (int64_t)0), g.ny - 1), iz = min(max(iz, (int64_t)0), g.nz - 1);
            c = (KeyT)((uint64_t)ix + (uint64_t)g.nx * ((uint64_t)iy + (uint64_t)g.ny * (uint64_t)iz));
        }
        code[i] = c, idx[i] = (uint32_t)i;
    }
}
void launch_cell_codes(const float4* pts, int64_t n, CellGrid g, uint64_t* code, uint32_t* idx, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_cell_codes<uint64_t>, dim3(grid_for(n)), dim3(kBlock), 0, s, pts, n, g, code, idx);
}
void launch_cell_codes32(const float4* pts, int64_t n, CellGrid g, uint32_t* code, uint32_t* idx, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_cell_codes<uint32_t>, dim3(grid_for(n)), dim3(kBlock), 0, s, pts, n, g, code, idx);
}

// Sorted copies of the points and the hash of the cell heads (one insert per occupied cell, linear probing).
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_cell_table(const float4* __restrict__ pts, const uint32_t* __restrict__ idx_sorted,
                                                       const KeyT* __restrict__ code_sorted, int64_t n, float4* __restrict__ pts_sorted,
                                                       CellHashEntry* __restrict__ table, uint32_t mask) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        pts_sorted[i] = pts[idx_sorted[i]];
        const KeyT c = code_sorted[i];
        if (c == (KeyT)~(KeyT)0) continue;  // non-finite points sort last and are never looked up
        if (i > 0 && code_sorted[i - 1] == c) continue;
        const uint64_t key = (uint64_t)c;
        uint32_t slot = (uint32_t)hash_cell(key) & mask;
        while (true) {
            const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(&table[slot].key), ~0ull, (unsigned long long)key);
            if (prev == ~0ull) {
                table[slot].start = (uint32_t)i;
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
}
void launch_cell_table(const float4* pts, const uint32_t* idx_sorted, const void* code_sorted, bool key32, int64_t n, float4* pts_sorted,
                       CellHashEntry* table, uint32_t table_mask, hipStream_t s) {
    if (n <= 0) return;
    if (key32)
        hipLaunchKernelGGL(k_cell_table<uint32_t>, dim3(grid_for(n)), dim3(kBlock), 0, s, pts, idx_sorted, (const uint32_t*)code_sorted, n, pts_sorted, table,
                           table_mask);
    else
        hipLaunchKernelGGL(k_cell_table<uint64_t>, dim3(grid_for(n)), dim3(kBlock), 0, s, pts, idx_sorted, (const uint64_t*)code_sorted, n, pts_sorted, table,
                           table_mask);
}

// offsets (dx + 1) + 3 (dy + 1) + 9 (dz + 1) of the 27 neighbour cells by increasing distance from the centre cell
__constant__ int kCellOrder[27] = {13, 4, 10, 12, 14, 16, 22, 1, 3, 5, 7, 9, 11, 15, 17, 19, 21, 23, 25, 0, 2, 6, 8, 18, 20, 24, 26};
// One thread per query: 27 cell lookups, then the exact float distance of flann::L2_Simple, ((0 + dx*dx) + dy*dy) + dz*dz,
// against the points of each cell (contiguous in the sorted copy); stops at the first hit.
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_radius_exists(const float4* __restrict__ query, int64_t nq, CellGrid g, const float4* __restrict__ pts_sorted,
                                                          const KeyT* __restrict__ code_sorted, int64_t n, const CellHashEntry* __restrict__ table,
                                                          uint32_t mask, float r2, uint8_t* __restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float4 q = query[i];
    int64_t cx, cy, cz;
    bool hit = false;
    if (n > 0 && cell_of(q, g, cx, cy, cz)) {
        // nearest cells first (own cell, 6 faces, 12 edges, 8 corners): most queries that have a neighbour at all find it in the
        // first one or two cells, and existence does not depend on the order
        for (int c = 0; c < 27 && !hit; ++c) {
            const int o = kCellOrder[c];
            const int64_t x = cx + (o % 3 - 1), y = cy + ((o / 3) % 3 - 1), z = cz + (o / 9 - 1);
            if (x < 0 || x >= g.nx || y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
            const uint64_t key = (uint64_t)x + (uint64_t)g.nx * ((uint64_t)y + (uint64_t)g.ny * (uint64_t)z);
            uint32_t slot = (uint32_t)hash_cell(key) & mask;
            int64_t start = -1;
            while (true) {
                const uint64_t k = table[slot].key;
                if (k == key) {
                    start = table[slot].start;
                    break;
                }
                if (k == ~0ull) break;
                slot = (slot + 1) & mask;
            }
            if (start < 0) continue;
            for (int64_t j = start; j < n && (uint64_t)code_sorted[j] == key; ++j) {
                const float4 p = pts_sorted[j];
                const float ddx = q.x - p.x, ddy = q.y - p.y, ddz = q.z - p.z;
                float d = 0.0f;
                d += ddx * ddx;
                d += ddy * ddy;
                d += ddz * ddz;
                if (d <= r2) {
                    hit = true;
                    break;
                }
            }
        }
    }
    flag[i] = hit ? 1 : 0;
}
// The same test with ONE WAVE per query: lanes 0 .. 26 look the 27 cells up side by side (one memory round trip instead of 27 dependent ones),
// then all 64 lanes walk the points of every occupied cell together -- coalesced 16-byte loads, one ballot per 64 candidates.  With a thread
// per query a few ten thousand queries leave most of the chip idle while single threads scan hundreds of points of dense cells one by one
// (addStaticPoints: 30 720 keyframe points against a 1.3 M-point window cloud, 590 us).  Existence does not depend on the order the
// candidates are looked at, the distance is the same float expression: the flags are the thread-per-query kernel's.
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_radius_exists_wave(const float4* __restrict__ query, int64_t nq, CellGrid g, const float4* __restrict__ pts_sorted,
                                                               const KeyT* __restrict__ code_sorted, int64_t n, const CellHashEntry* __restrict__ table,
                                                               uint32_t mask, float r2, uint8_t* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (i >= nq) return;  // (whole waves)
    const float4 q = query[i];
    int64_t cx, cy, cz;
    bool hit = false;
    if (n > 0 && cell_of(q, g, cx, cy, cz)) {
        int64_t start = -1;
        uint64_t key = 0;
        if (lane < 27) {
            const int o = kCellOrder[lane];
            const int64_t x = cx + (o % 3 - 1), y = cy + ((o / 3) % 3 - 1), z = cz + (o / 9 - 1);
            if (!(x < 0 || x >= g.nx || y < 0 || y >= g.ny || z < 0 || z >= g.nz)) {
                key = (uint64_t)x + (uint64_t)g.nx * ((uint64_t)y + (uint64_t)g.ny * (uint64_t)z);
                uint32_t slot = (uint32_t)hash_cell(key) & mask;
                while (true) {
                    const uint64_t k = table[slot].key;
                    if (k == key) {
                        start = table[slot].start;
                        break;
                    }
                    if (k == ~0ull) break;
                    slot = (slot + 1) & mask;
                }
            }
        }
        unsigned long long occupied = __ballot(start >= 0);
        while (occupied != 0ull && !hit) {
            const int c = __builtin_ctzll(occupied);
            occupied &= occupied - 1ull;
            const int64_t st = __shfl(start, c);
            const uint64_t kc = __shfl(key, c);
            for (int64_t j0 = st; j0 < n; j0 += 64) {
                const int64_t j = j0 + lane;
                const bool in_cell = j < n && (uint64_t)code_sorted[j] == kc;
                bool near = false;
                if (in_cell) {
                    const float4 p = pts_sorted[j];
                    const float ddx = q.x - p.x, ddy = q.y - p.y, ddz = q.z - p.z;
                    float d = 0.0f;
                    d += ddx * ddx;
                    d += ddy * ddy;
                    d += ddz * ddz;
                    near = d <= r2;
                }
                if (__ballot(near) != 0ull) {
                    hit = true;
                    break;
                }
                if (__ballot(in_cell) != ~0ull) break;  // the cell's run ended inside these 64
            }
        }
    }
    if (lane == 0) flag[i] = hit ? 1 : 0;
}
void launch_radius_exists(const float4* query, int64_t nq, CellGrid g, const float4* pts_sorted, const void* code_sorted, bool key32, int64_t n,
                          const CellHashEntry* table, uint32_t table_mask, float r2, uint8_t* flag, hipStream_t s) {
    if (nq <= 0) return;
    // A wave per query costs ~0.4 ns per query whatever the cells hold; a thread per query costs as much as its slowest thread, which walks
    // the points of up to 27 cells one by one.  Measured (scripts/static_time.py): 30 720 queries against the 1.3 M-point window cloud 323 ->
    // 27 us with waves; 1.3 M queries against 18 210 static points 72 us with threads, 550 us with waves.
    if (nq <= ((int64_t)1 << 17) || n >= 4 * nq) {
        const unsigned wgrid = (unsigned)((nq * 64 + kBlock - 1) / kBlock);
        if (key32)
            hipLaunchKernelGGL(k_radius_exists_wave<uint32_t>, dim3(wgrid), dim3(kBlock), 0, s, query, nq, g, pts_sorted, (const uint32_t*)code_sorted, n, table,
                               table_mask, r2, flag);
        else
            hipLaunchKernelGGL(k_radius_exists_wave<uint64_t>, dim3(wgrid), dim3(kBlock), 0, s, query, nq, g, pts_sorted, (const uint64_t*)code_sorted, n, table,
                               table_mask, r2, flag);
        return;
    }
    const unsigned grid = (unsigned)((nq + kBlock - 1) / kBlock);
    if (key32)
        hipLaunchKernelGGL(k_radius_exists<uint32_t>, dim3(grid), dim3(kBlock), 0, s, query, nq, g, pts_sorted, (const uint32_t*)code_sorted, n, table, table_mask,
                           r2, flag);
    else
        hipLaunchKernelGGL(k_radius_exists<uint64_t>, dim3(grid), dim3(kBlock), 0, s, query, nq, g, pts_sorted, (const uint64_t*)code_sorted, n, table, table_mask,
                           r2, flag);
}

// isVisible (DmsaSlam.h:360-375): d = p.n ; res = pos.n - d ; visible iff res >= -0.00001 (double compare); 3-term inner
// products as x0 + (x1 + x2) like every fixed-size Eigen product on the path.
__global__ __launch_bounds__(kBlock) void k_static_flags(const float4* __restrict__ key_xyz, const float4* __restrict__ key_normal,
                                                         const uint8_t* __restrict__ within, int64_t n, float px, float py, float pz,
                                                         int32_t* __restrict__ sel) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s = 0;
    if (within[i]) {
        const float4 p = key_xyz[i], nn = key_normal[i];
        const float d = p.x * nn.x + (p.y * nn.y + p.z * nn.z);
        const float res = (px * nn.x + (py * nn.y + pz * nn.z)) - d;
        s = ((double)res >= -0.00001) ? 1 : 0;
    }
    sel[i] = s;
}
void launch_static_flags(const float4* key_xyz, const float4* key_normal, const uint8_t* within, int64_t n, float px, float py, float pz, int32_t* sel,
                         hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_static_flags, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, key_xyz, key_normal, within, n, px, py, pz, sel);
}
__global__ __launch_bounds__(kBlock) void k_static_scatter(const float4* __restrict__ key_xyz, const int32_t* __restrict__ key_ring,
                                                           const int32_t* __restrict__ sel, const int32_t* __restrict__ scan_excl, int64_t n,
                                                           float4* __restrict__ out_xyz, int32_t* __restrict__ out_id) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !sel[i]) return;
    const float4 p = key_xyz[i];
    const int o = scan_excl[i];
    out_xyz[o] = make_float4(p.x, p.y, p.z, 1.0f);
    out_id[o] = key_ring[i];
}
void launch_static_scatter(const float4* key_xyz, const int32_t* key_ring, const int32_t* sel, const int32_t* scan_excl, int64_t n, float4* out_xyz,
                           int32_t* out_id, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_static_scatter, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, key_xyz, key_ring, sel, scan_excl, n, out_xyz, out_id);
}
__global__ void k_pick_offsets(const int32_t* __restrict__ scan_excl, const int32_t* __restrict__ sel, const int64_t* __restrict__ offsets, int K,
                               int64_t n, int32_t* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > K) return;
    const int64_t o = offsets[k];
    out[k] = n == 0 ? 0 : (o < n ? scan_excl[o] : scan_excl[n - 1] + sel[n - 1]);
}
void launch_pick_offsets(const int32_t* scan_excl, const int32_t* sel, const int64_t* offsets, int K, int64_t n, int32_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_pick_offsets, dim3((unsigned)((K + 1 + 63) / 64)), dim3(64), 0, s, scan_excl, sel, offsets, K, n, out);
}
// flags are bytes of value 0 or 1 (k_radius_exists): sixteen of them per load, summed as packed bytes; one atomic per workgroup
__global__ __launch_bounds__(kBlock) void k_count_flags(const uint8_t* __restrict__ flag, int64_t n, unsigned long long* __restrict__ count) {
    __shared__ unsigned s_c[kBlock / 64];
    unsigned c = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, n16 = n >> 4;
    const uint4* f16 = reinterpret_cast<const uint4*>(flag);  // (hipMalloc'd: 256-byte aligned)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 w = f16[i];
        const unsigned t = (w.x & 0x01010101u) + (w.y & 0x01010101u) + (w.z & 0x01010101u) + (w.w & 0x01010101u);  // four byte lanes, each <= 4
        c += (t * 0x01010101u) >> 24;
    }
    for (int64_t i = (n16 << 4) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) c += flag[i] & 1u;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += (unsigned)__shfl_xor((int)c, m);
    if ((threadIdx.x & 63) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += s_c[w];
        if (t) atomicAdd(count, (unsigned long long)t);
    }
}
void launch_count_flags(const uint8_t* flag, int64_t n, unsigned long long* count, hipStream_t s) {
    (void)hipMemsetAsync(count, 0, sizeof(unsigned long long), s);
    if (n > 0) hipLaunchKernelGGL(k_count_flags, dim3(grid_for((n + 15) / 16, kBlock, 256)), dim3(kBlock), 0, s, flag, n, count);
}
__global__ __launch_bounds__(kBlock) void k_leaf_pick(const int32_t* __restrict__ leaf_start, const uint32_t* __restrict__ idx_sorted,
                                                      const int32_t* __restrict__ rnd, int num_leaves, int32_t* __restrict__ out) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= num_leaves) return;
    const int b = leaf_start[l], cnt = leaf_start[l + 1] - b;
    const double r = (double)rnd[l] / 2147483647.0;         // (double)rand() / RAND_MAX
    const int id = (int)(r * (double)(cnt - 1));            // static_cast<int>(r * (double)(indices.size() - 1))
    out[l] = (int32_t)idx_sorted[b + id];
}
void launch_leaf_pick(const int32_t* leaf_start, const uint32_t* idx_sorted, const int32_t* rnd, int num_leaves, int32_t* out, hipStream_t s) {
    if (num_leaves > 0) hipLaunchKernelGGL(k_leaf_pick, dim3((unsigned)((num_leaves + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, leaf_start, idx_sorted, rnd, num_leaves, out);
}

// ---- DmsaSlam::preProcess (DmsaSlam.h:594-630) after the grid filter ----------------------------------------------------------
// ranges[k] = Vector3f(x, y, z).norm() of filtered point k (:601): sqrt(x0^2 + (x1^2 + x2^2)), correctly rounded float sqrt.
// Ranges are non-negative (or +inf), so their bit patterns sort like the floats.
__global__ __launch_bounds__(kBlock) void k_scan_ranges(const float4* __restrict__ raw, const int32_t* __restrict__ pick, int m,
                                                        uint32_t* __restrict__ range_bits, uint32_t* __restrict__ iota) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const float4 p = raw[pick[k]];
    const float xx = p.x * p.x, yy = p.y * p.y, zz = p.z * p.z;
    const float t = yy + zz;
    range_bits[k] = __float_as_uint(sqrtf(xx + t));
    iota[k] = (uint32_t)k;
}
// :609 thresRange = std::max(rangesSorted[min(max_num_points_per_scan, size - 1)], minDistDS); :616 keep iff range < thresRange &&
// range > min_dist.  The threshold is read from the sorted ranges on the device: no host round trip between sort and gate.
__global__ __launch_bounds__(kBlock) void k_scan_range_gate(const uint32_t* __restrict__ range_bits, const uint32_t* __restrict__ sorted_bits, int m,
                                                            int thres_pos, float min_dist_ds, float min_dist, int32_t* __restrict__ sel) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const float a = __uint_as_float(sorted_bits[thres_pos]);
    const float thres = a < min_dist_ds ? min_dist_ds : a;  // std::max(a, b) = (a < b) ? b : a
    const float r = __uint_as_float(range_bits[k]);
    sel[k] = (r < thres && r > min_dist) ? 1 : 0;
}
// stable compaction (pcl::ExtractIndices keeps the order) + pcl::transformPointCloud(lidarToImuTform) + data[3] = 1 (:620-630).
// Per component x*c0 + (y*c1 + (z*c2 + c3)): pcl::detail::Transformer<float>::se3 of PCL >= 1.10 (SSE2 build), no FMA.
struct Mat4ColMajor { float m[16]; };
__global__ __launch_bounds__(kBlock) void k_scan_emit(const float4* __restrict__ raw, const int32_t* __restrict__ pick, const int32_t* __restrict__ sel,
                                                      const int32_t* __restrict__ scan_excl, int m, Mat4ColMajor T, float4* __restrict__ out_xyz,
                                                      int32_t* __restrict__ out_src, int32_t* __restrict__ total) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    if (k == m - 1) *total = scan_excl[k] + sel[k];
    if (!sel[k]) return;
    const int src = pick[k];
    const float4 p = raw[src];
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = p.x * T.m[c], b = p.y * T.m[4 + c], d = p.z * T.m[8 + c];
        const float t2 = d + T.m[12 + c];
        const float t1 = b + t2;
        o[c] = a + t1;
    }
    const int at = scan_excl[k];
    out_xyz[at] = make_float4(o[0], o[1], o[2], 1.0f);
    out_src[at] = src;
}
void launch_scan_ranges(const float4* raw, const int32_t* pick, int m, uint32_t* range_bits, uint32_t* iota, hipStream_t s) {
    if (m > 0) hipLaunchKernelGGL(k_scan_ranges, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, raw, pick, m, range_bits, iota);
}
void launch_scan_range_gate(const uint32_t* range_bits, const uint32_t* sorted_bits, int m, int thres_pos, float min_dist_ds, float min_dist, int32_t* sel,
                            hipStream_t s) {
    if (m > 0)
        hipLaunchKernelGGL(k_scan_range_gate, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, range_bits, sorted_bits, m, thres_pos, min_dist_ds,
                           min_dist, sel);
}
void launch_scan_emit(const float4* raw, const int32_t* pick, const int32_t* sel, const int32_t* scan_excl, int m, const float* tform_colmajor, float4* out_xyz,
                      int32_t* out_src, int32_t* total, hipStream_t s) {
    Mat4ColMajor T;
    for (int i = 0; i < 16; ++i) T.m[i] = tform_colmajor[i];
    if (m > 0)
        hipLaunchKernelGGL(k_scan_emit, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, raw, pick, sel, scan_excl, m, T, out_xyz, out_src, total);
}

// ---- ContinuousTrajectory::registerPcBuffer (ContinuousTrajectory.h:240-260) ------------------------------------------------------
// tformIdPerPoint[k] = min(lower_bound(trajTime, stamp_k - t0), n_total - 1): std::lower_bound's own bisection, one per point, on
// the dense time grid staged in LDS (8 KB at n_total = 1002).  A NaN stamp compares false everywhere and lands on index 0, like the
// reference.
constexpr int kTimeGridLds = 8192;  // doubles: 64 KB
__global__ __launch_bounds__(kBlock) void k_tform_indices(const double* __restrict__ stamps, int64_t n, double t0, const double* __restrict__ traj_time,
                                                          int n_total, int use_lds, int32_t* __restrict__ out) {
    extern __shared__ double s_time[];
    const double* grid = traj_time;
    if (use_lds) {
        for (int i = threadIdx.x; i < n_total; i += blockDim.x) s_time[i] = traj_time[i];
        __syncthreads();
        grid = s_time;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = stamps[i] - t0;
        int first = 0, len = n_total;
        while (len > 0) {
            const int half = len >> 1, mid = first + half;
            if (grid[mid] < v)
                first = mid + 1, len = len - half - 1;
            else
                len = half;
        }
        out[i] = first < n_total - 1 ? first : n_total - 1;
    }
}
// One scan of the resident window ring into the window's point arrays: row = min(lower_bound(trajTime, stamp - t0), n_total - 1) like
// k_tform_indices, packed into the w of the local point; ring ids copied along.
__global__ __launch_bounds__(kBlock) void k_ring_assemble(const float4* __restrict__ xyz, const double* __restrict__ stamps, const int32_t* __restrict__ ring,
                                                          int64_t n, double t0, const double* __restrict__ traj_time, int n_total, int use_lds,
                                                          float4* __restrict__ local_out, int32_t* __restrict__ ring_out) {
    extern __shared__ double s_time[];
    const double* grid = traj_time;
    if (use_lds) {
        for (int i = threadIdx.x; i < n_total; i += blockDim.x) s_time[i] = traj_time[i];
        __syncthreads();
        grid = s_time;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = stamps[i] - t0;
        int first = 0, len = n_total;
        while (len > 0) {
            const int half = len >> 1, mid = first + half;
            if (grid[mid] < v)
                first = mid + 1, len = len - half - 1;
            else
                len = half;
        }
        const int row = first < n_total - 1 ? first : n_total - 1;
        const float4 q = xyz[i];
        local_out[i] = make_float4(q.x, q.y, q.z, __int_as_float(row));
        ring_out[i] = ring[i];
    }
}
void launch_ring_assemble(const float4* xyz, const double* stamps, const int32_t* ring, int64_t n, double t0, const double* traj_time, int n_total,
                          float4* local_out, int32_t* ring_out, hipStream_t s) {
    if (n <= 0) return;
    const int use_lds = n_total <= kTimeGridLds ? 1 : 0;
    hipLaunchKernelGGL(k_ring_assemble, dim3(grid_for(n, kBlock, 2048)), dim3(kBlock), use_lds ? (size_t)n_total * sizeof(double) : 0, s, xyz, stamps, ring, n, t0,
                       traj_time, n_total, use_lds, local_out, ring_out);
}
void launch_tform_indices(const double* stamps, int64_t n, double t0, const double* traj_time, int n_total, int32_t* out, hipStream_t s) {
    if (n <= 0) return;
    const int use_lds = n_total <= kTimeGridLds ? 1 : 0;
    hipLaunchKernelGGL(k_tform_indices, dim3(grid_for(n, kBlock, 2048)), dim3(kBlock), use_lds ? (size_t)n_total * sizeof(double) : 0, s, stamps, n, t0, traj_time,
                       n_total, use_lds, out);
}

// ---- dmsa_slam_ros::callbackPointCloud (src/dmsa_slam_ros.cpp:399-486) -------------------------------------------------------------
// One point per thread: memcpy-style reads from the message blob at the field offsets the sensor type names.  The blob is read
// byte-wise (fields are not aligned in general); a 131 072-point scan is 4-6 MB, the L2 absorbs the overlap between neighbours.
template <typename T>
__device__ __forceinline__ T load_unaligned(const uint8_t* p) {
    T v;
    uint8_t* d = reinterpret_cast<uint8_t*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T); ++i) d[i] = p[i];
    return v;
}
__global__ __launch_bounds__(kBlock) void k_decode_pointcloud2(const uint8_t* __restrict__ data, uint32_t n, uint32_t point_step, PointCloud2Fields f, int sensor,
                                                               double stamp_msg, double delta_t, float4* __restrict__ xyz, double* __restrict__ stamp,
                                                               int32_t* __restrict__ id) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint8_t* p = data + (size_t)(k * point_step);  // uint32 product like the reference's `k * msg->point_step`
    xyz[k] = make_float4(load_unaligned<float>(p + f.x), load_unaligned<float>(p + f.y), load_unaligned<float>(p + f.z), 0.0f);
    double st;
    int32_t ring;
    switch (sensor) {
        case 0: st = load_unaligned<double>(p + f.stamp), ring = (int32_t)load_unaligned<uint16_t>(p + f.ring); break;                                   // hesai
        case 1: st = stamp_msg + 1e-9 * (double)load_unaligned<uint32_t>(p + f.stamp), ring = (int32_t)load_unaligned<uint8_t>(p + f.ring); break;      // ouster
        case 2: st = load_unaligned<double>(p + f.stamp), ring = (int32_t)load_unaligned<uint16_t>(p + f.ring); break;                                   // robosense
        case 3: st = stamp_msg + (double)load_unaligned<float>(p + f.stamp), ring = (int32_t)load_unaligned<uint16_t>(p + f.ring); break;                // velodyne
        case 4: st = load_unaligned<double>(p + f.stamp), ring = (int32_t)(k % 1000u); break;                                                            // livox, seconds
        case 5: st = 1e-9 * load_unaligned<double>(p + f.stamp), ring = (int32_t)(k % 1000u); break;                                                     // livox, nanoseconds
        case 6: st = stamp_msg + (double)load_unaligned<float>(p + f.stamp), ring = (int32_t)load_unaligned<int8_t>(p + f.ring); break;                  // sick
        default: st = stamp_msg + delta_t * (double)k / (double)n, ring = (int32_t)(k % 1000u); break;                                                   // unknown
    }
    stamp[k] = st;
    id[k] = ring;
}
void launch_decode_pointcloud2(const uint8_t* data, uint32_t n, uint32_t point_step, PointCloud2Fields f, int sensor, double stamp_msg, double delta_t, float4* xyz,
                               double* stamp, int32_t* id, hipStream_t s) {
    if (n > 0)
        hipLaunchKernelGGL(k_decode_pointcloud2, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, data, n, point_step, f, sensor, stamp_msg, delta_t, xyz, stamp, id);
}

// ---- DmsaSlam::updateNormals (DmsaSlam.h:553-567): pcl::NormalEstimationOMP, k nearest neighbours, viewpoint flip ---------------------
// Exact k-NN on the sorted cell grid: rings of cells around the query's cell are scanned until the k-th best distance lies strictly
// inside the radius the scanned block guarantees (ring * cell size), or the block covers the whole grid.  Candidates are ordered by
// (flann::L2_Simple float distance, point index) -- the order of a sorted FLANN result with ties broken by index.
constexpr int kMaxNeighbours = 8;
constexpr int kMaxRings = 6;  // (2 * 6 + 1)^3 = 2197 cells before an isolated query falls back to an exhaustive pass
struct NeighbourList {
    float d[kMaxNeighbours];
    uint32_t pos[kMaxNeighbours];  // position in the sorted copy
    int count;
};
template <int K>
__device__ __forceinline__ void knn_insert(NeighbourList& nl, int k, float d, uint32_t pos, const uint32_t* __restrict__ idx_sorted) {
    // every index below is a compile-time constant after unrolling: the lists stay in registers, no indirect register addressing
    if (nl.count == k) {
        float last_d = 0.0f;
        uint32_t last_pos = 0;
#pragma unroll
        for (int m = 0; m < K; ++m)
            if (m == k - 1) last_d = nl.d[m], last_pos = nl.pos[m];
        if (d > last_d) return;
        if (d == last_d && idx_sorted[pos] > idx_sorted[last_pos]) return;
    }
    float cd = d;
    uint32_t cp = pos;
    bool placed = false;
#pragma unroll
    for (int m = 0; m < K; ++m) {
        if (m < k && !placed) {
            if (m >= nl.count) {
                nl.d[m] = cd, nl.pos[m] = cp, placed = true;
            } else {
                const bool before = cd < nl.d[m] || (cd == nl.d[m] && idx_sorted[cp] < idx_sorted[nl.pos[m]]);
                if (before) {
                    const float td = nl.d[m];
                    const uint32_t tp = nl.pos[m];
                    nl.d[m] = cd, nl.pos[m] = cp;
                    cd = td, cp = tp;
                }
            }
        }
    }
    if (nl.count < k) nl.count += 1;
}
template <typename KeyT>
__device__ __forceinline__ void knn_scan_cell(NeighbourList& nl, int k, const float4 q, int64_t x, int64_t y, int64_t z, const CellGrid& g,
                                              const float4* __restrict__ pts_sorted, const uint32_t* __restrict__ idx_sorted, const KeyT* __restrict__ code_sorted,
                                              int64_t n, const CellHashEntry* __restrict__ table, uint32_t mask) {
    const uint64_t key = (uint64_t)x + (uint64_t)g.nx * ((uint64_t)y + (uint64_t)g.ny * (uint64_t)z);
    uint32_t slot = (uint32_t)hash_cell(key) & mask;
    int64_t start = -1;
    while (true) {
        const uint64_t kk = table[slot].key;
        if (kk == key) {
            start = table[slot].start;
            break;
        }
        if (kk == ~0ull) break;
        slot = (slot + 1) & mask;
    }
    if (start < 0) return;
    for (int64_t j = start; j < n && (uint64_t)code_sorted[j] == key; ++j) {
        const float4 p = pts_sorted[j];
        const float ddx = q.x - p.x, ddy = q.y - p.y, ddz = q.z - p.z;
        float d = 0.0f;
        d += ddx * ddx;
        d += ddy * ddy;
        d += ddz * ddz;
        knn_insert<kMaxNeighbours>(nl, k, d, (uint32_t)j, idx_sorted);
    }
}
// pcl::computeRoots2 / computeRoots / eigen33 (pcl/common/impl/eigen.hpp, PCL 1.10), smallest eigenvalue and its vector, float
__device__ __forceinline__ void pcl_roots2(float b, float c, float* roots) {
    roots[0] = 0.0f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);  // Scalar (b * b - 4.0 * c): the subtraction runs in double
    if (d < 0.0f) d = 0.0f;
    const float sd = sqrtf(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}
__device__ __forceinline__ void pcl_roots(const float* m /* row-major 3x3 */, float* roots) {
    const float m00 = m[0], m01 = m[1], m02 = m[2], m11 = m[4], m12 = m[5], m22 = m[8];
    const float c0 = m00 * m11 * m22 + 2.0f * m01 * m02 * m12 - m00 * m12 * m12 - m11 * m02 * m02 - m22 * m01 * m01;
    const float c1 = m00 * m11 - m01 * m01 + m00 * m22 - m02 * m02 + m11 * m22 - m12 * m12;
    const float c2 = m00 + m11 + m22;
    if (fabsf(c0) < FLT_EPSILON) {
        pcl_roots2(c2, c1, roots);
        return;
    }
    const float s_inv3 = (float)(1.0 / 3.0), s_sqrt3 = sqrtf(3.0f);
    const float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    const float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    const float rho = sqrtf(-a_over_3);
    // float atan2 / cos / sin as a correctly rounded libm returns them: evaluated in double, rounded once (glibc's sinf / cosf work
    // the same way; device and host double functions agree after the rounding, so normals are reproducible across the two)
    const float theta = (float)dmsa_det::det_atan2((double)sqrtf(-q), (double)half_b) * s_inv3;
    const float cos_theta = (float)dmsa_det::det_cos((double)theta), sin_theta = (float)dmsa_det::det_sin((double)theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) t = roots[0], roots[0] = roots[1], roots[1] = t;
    if (roots[1] >= roots[2]) {
        t = roots[1], roots[1] = roots[2], roots[2] = t;
        if (roots[0] >= roots[1]) t = roots[0], roots[0] = roots[1], roots[1] = t;
    }
    if (roots[0] <= 0.0f) pcl_roots2(c2, c1, roots);
}
template <typename KeyT>
__global__ __launch_bounds__(kBlock) void k_knn_normals(const float4* __restrict__ cloud, int64_t n, int k, CellGrid g, double cell_size,
                                                        const float4* __restrict__ pts_sorted, const uint32_t* __restrict__ idx_sorted,
                                                        const KeyT* __restrict__ code_sorted, const CellHashEntry* __restrict__ table, uint32_t mask,
                                                        uint32_t num_finite, float vpx, float vpy, float vpz, float4* __restrict__ normal,
                                                        int32_t* __restrict__ nn_index) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = cloud[i];
    const float nanv = __int_as_float(0x7fc00000);
    NeighbourList nl;
    nl.count = 0;
    int64_t cx, cy, cz;
    const int want = (int)min((uint32_t)k, num_finite);
    if (want > 0 && cell_of(q, g, cx, cy, cz)) {
        cx = min(max(cx, (int64_t)0), g.nx - 1), cy = min(max(cy, (int64_t)0), g.ny - 1), cz = min(max(cz, (int64_t)0), g.nz - 1);
        const int64_t reach = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));  // ring that covers the whole grid
        for (int64_t ring = 0; ring <= reach; ++ring) {
            if (ring > kMaxRings) {
                // an isolated query (its neighbours are dozens of cells away): the shells grow as ring^2 hash probes, an exhaustive pass
                // over the cloud is cheaper.  Start over so that nothing is inserted twice.
                nl.count = 0;
                for (int64_t j = 0; j < n; ++j) {
                    if (code_sorted[j] == (KeyT)~(KeyT)0) break;  // non-finite points sort last
                    const float4 p = pts_sorted[j];
                    const float ddx = q.x - p.x, ddy = q.y - p.y, ddz = q.z - p.z;
                    float d = 0.0f;
                    d += ddx * ddx;
                    d += ddy * ddy;
                    d += ddz * ddz;
                    knn_insert<kMaxNeighbours>(nl, want, d, (uint32_t)j, idx_sorted);
                }
                break;
            }
            for (int64_t dz = -ring; dz <= ring; ++dz) {
                const int64_t z = cz + dz;
                if (z < 0 || z >= g.nz) continue;
                for (int64_t dy = -ring; dy <= ring; ++dy) {
                    const int64_t y = cy + dy;
                    if (y < 0 || y >= g.ny) continue;
                    const bool face = dz == -ring || dz == ring || dy == -ring || dy == ring;
                    for (int64_t dx = -ring; dx <= ring; dx += (face || ring == 0) ? 1 : 2 * ring) {  // only the shell of the block
                        const int64_t x = cx + dx;
                        if (x < 0 || x >= g.nx) continue;
                        knn_scan_cell<KeyT>(nl, want, q, x, y, z, g, pts_sorted, idx_sorted, code_sorted, n, table, mask);
                    }
                }
            }
            // every point outside the scanned block is at least ring * cell away (the clamp above only moves queries that sit on the
            // upper faces of the grid, where nothing lies beyond); 0.999 absorbs the rounding of the cell assignment
            const double safe = 0.999 * (double)ring * cell_size;
            float worst = 0.0f;
#pragma unroll
            for (int m = 0; m < kMaxNeighbours; ++m)
                if (m == want - 1) worst = nl.d[m];
            if (nl.count == want && (double)worst < safe * safe) break;
        }
    }
    if (nn_index) {
#pragma unroll
        for (int m = 0; m < kMaxNeighbours; ++m)
            if (m < k) nn_index[i * k + m] = m < nl.count ? (int32_t)idx_sorted[nl.pos[m]] : -1;
    }
    if (nl.count < 3) {  // NormalEstimation: non-finite query, or pcl::computePointNormal with fewer than 3 indices
        normal[i] = make_float4(nanv, nanv, nanv, nanv);
        return;
    }
    // pcl::computeMeanAndCovarianceMatrix (single pass, float accumulators, neighbours in search order)
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
#pragma unroll
    for (int m = 0; m < kMaxNeighbours; ++m) {
        if (m >= nl.count) continue;
        const float4 p = pts_sorted[nl.pos[m]];
        a0 += p.x * p.x, a1 += p.x * p.y, a2 += p.x * p.z, a3 += p.y * p.y, a4 += p.y * p.z, a5 += p.z * p.z, a6 += p.x, a7 += p.y, a8 += p.z;
    }
    const float cnt = (float)nl.count;
    a0 /= cnt, a1 /= cnt, a2 /= cnt, a3 /= cnt, a4 /= cnt, a5 /= cnt, a6 /= cnt, a7 /= cnt, a8 /= cnt;
    float cov[9];
    cov[0] = a0 - a6 * a6, cov[1] = a1 - a6 * a7, cov[2] = a2 - a6 * a8, cov[4] = a3 - a7 * a7, cov[5] = a4 - a7 * a8, cov[8] = a5 - a8 * a8;
    cov[3] = cov[1], cov[6] = cov[2], cov[7] = cov[5];
    // pcl::solvePlaneParameters -> pcl::eigen33 (smallest eigenvalue)
    float scale = 0.0f;
#pragma unroll
    for (int e = 0; e < 9; ++e) scale = fmaxf(scale, fabsf(cov[e]));
    if (scale <= FLT_MIN) scale = 1.0f;
    float sm[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) sm[e] = cov[e] / scale;
    float roots[3];
    pcl_roots(sm, roots);
    const float eigenvalue = roots[0] * scale;
    sm[0] -= roots[0], sm[4] -= roots[0], sm[8] -= roots[0];
    const float r0x = sm[0], r0y = sm[1], r0z = sm[2], r1x = sm[3], r1y = sm[4], r1z = sm[5], r2x = sm[6], r2y = sm[7], r2z = sm[8];
    const float v1x = r0y * r1z - r0z * r1y, v1y = r0z * r1x - r0x * r1z, v1z = r0x * r1y - r0y * r1x;
    const float v2x = r0y * r2z - r0z * r2y, v2y = r0z * r2x - r0x * r2z, v2z = r0x * r2y - r0y * r2x;
    const float v3x = r1y * r2z - r1z * r2y, v3y = r1z * r2x - r1x * r2z, v3z = r1x * r2y - r1y * r2x;
    const float len1 = v1x * v1x + (v1y * v1y + v1z * v1z), len2 = v2x * v2x + (v2y * v2y + v2z * v2z), len3 = v3x * v3x + (v3y * v3y + v3z * v3z);
    float nx, ny, nz;
    if (len1 >= len2 && len1 >= len3) {
        const float s = sqrtf(len1);
        nx = v1x / s, ny = v1y / s, nz = v1z / s;
    } else if (len2 >= len1 && len2 >= len3) {
        const float s = sqrtf(len2);
        nx = v2x / s, ny = v2y / s, nz = v2z / s;
    } else {
        const float s = sqrtf(len3);
        nx = v3x / s, ny = v3y / s, nz = v3z / s;
    }
    const float eig_sum = cov[0] + cov[4] + cov[8];
    const float curvature = eig_sum != 0.0f ? fabsf(eigenvalue / eig_sum) : 0.0f;
    // pcl::flipNormalTowardsViewpoint
    const float wx = vpx - q.x, wy = vpy - q.y, wz = vpz - q.z;
    const float cos_theta = wx * nx + wy * ny + wz * nz;
    if (cos_theta < 0.0f) nx *= -1.0f, ny *= -1.0f, nz *= -1.0f;
    normal[i] = make_float4(nx, ny, nz, curvature);
}
void launch_knn_normals(const float4* cloud, int64_t n, int k, CellGrid g, double cell_size, const float4* pts_sorted, const uint32_t* idx_sorted,
                        const void* code_sorted, bool key32, const CellHashEntry* table, uint32_t table_mask, uint32_t num_finite, float vpx, float vpy, float vpz,
                        float4* normal, int32_t* nn_index, hipStream_t s) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    if (key32)
        hipLaunchKernelGGL(k_knn_normals<uint32_t>, dim3(grid), dim3(kBlock), 0, s, cloud, n, k, g, cell_size, pts_sorted, idx_sorted, (const uint32_t*)code_sorted,
                           table, table_mask, num_finite, vpx, vpy, vpz, normal, nn_index);
    else
        hipLaunchKernelGGL(k_knn_normals<uint64_t>, dim3(grid), dim3(kBlock), 0, s, cloud, n, k, g, cell_size, pts_sorted, idx_sorted, (const uint64_t*)code_sorted,
                           table, table_mask, num_finite, vpx, vpy, vpz, normal, nn_index);
}

// addNewKeyframeToMap (DmsaSlam.h:518-523): p_local = currRotInv * (p_global - currWorldPose), float, 3-term products as x0 + (x1 + x2)
struct Mat3f { float m[9]; };  // row-major
__global__ __launch_bounds__(kBlock) void k_to_keyframe_frame(const float4* __restrict__ global, const int32_t* __restrict__ ids, const int32_t* __restrict__ pick,
                                                              int m, Mat3f Rinv, float tx, float ty, float tz, float4* __restrict__ local, int32_t* __restrict__ ring) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const int src = pick[k];
    const float4 p = global[src];
    const float dx = p.x - tx, dy = p.y - ty, dz = p.z - tz;
    local[k] = make_float4(Rinv.m[0] * dx + (Rinv.m[1] * dy + Rinv.m[2] * dz), Rinv.m[3] * dx + (Rinv.m[4] * dy + Rinv.m[5] * dz),
                           Rinv.m[6] * dx + (Rinv.m[7] * dy + Rinv.m[8] * dz), 1.0f);
    ring[k] = ids[src];
}
void launch_to_keyframe_frame(const float4* global, const int32_t* ids, const int32_t* pick, int m, const float* rinv_rowmajor, float tx, float ty, float tz,
                              float4* local, int32_t* ring, hipStream_t s) {
    Mat3f R;
    for (int i = 0; i < 9; ++i) R.m[i] = rinv_rowmajor[i];
    if (m > 0) hipLaunchKernelGGL(k_to_keyframe_frame, dim3((unsigned)((m + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, global, ids, pick, m, R, tx, ty, tz, local, ring);
}

// ---- include/dmsa_aos.h: the caller's strided clouds, packed on the device ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_pack_aos_window(const uint8_t* __restrict__ raw, int64_t count, int stride, int xyz_offset, int aux_offset,
                                                            const int32_t* __restrict__ index, int row_limit, int fixed_row, float4* __restrict__ local_out,
                                                            int32_t* __restrict__ ring_out, int32_t* __restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint8_t* pt = raw + (size_t)i * stride;
    const float* xyz = reinterpret_cast<const float*>(pt + xyz_offset);  // stride and offsets are multiples of four
    int row = fixed_row;
    if (index != nullptr) {
        row = index[i];
        if (row < 0 || row >= row_limit) {
            *bad = 1;
            row = 0;
        }
    }
    local_out[i] = make_float4(xyz[0], xyz[1], xyz[2], __int_as_float(row));
    ring_out[i] = *reinterpret_cast<const int32_t*>(pt + aux_offset);
}
__global__ __launch_bounds__(kBlock) void k_pack_aos_keyframe(const uint8_t* __restrict__ raw, int64_t count, int stride, int xyz_offset, int aux_offset, int row,
                                                              float4* __restrict__ local_out, float4* __restrict__ normal_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint8_t* pt = raw + (size_t)i * stride;
    const float* xyz = reinterpret_cast<const float*>(pt + xyz_offset);
    const float* nrm = reinterpret_cast<const float*>(pt + aux_offset);
    local_out[i] = make_float4(xyz[0], xyz[1], xyz[2], __int_as_float(row));
    normal_out[i] = make_float4(nrm[0], nrm[1], nrm[2], nrm[3]);
}
// the scan of a window as the ring keeps it (include/dmsa_window_ring.h): coordinates, stamps, ring ids out of a strided cloud
__global__ __launch_bounds__(kBlock) void k_unpack_ring_scan(const uint8_t* __restrict__ raw, int64_t count, int stride, int xyz_offset, int stamp_offset, int id_offset,
                                                             float4* __restrict__ xyz_out, double* __restrict__ stamp_out, int32_t* __restrict__ id_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint8_t* pt = raw + (size_t)i * stride;
    const float* xyz = reinterpret_cast<const float*>(pt + xyz_offset);
    xyz_out[i] = make_float4(xyz[0], xyz[1], xyz[2], 1.0f);
    const uint32_t* sw = reinterpret_cast<const uint32_t*>(pt + stamp_offset);  // (4-byte aligned reads: the stride need not be a multiple of eight)
    stamp_out[i] = __longlong_as_double((long long)(((unsigned long long)sw[1] << 32) | sw[0]));
    id_out[i] = *reinterpret_cast<const int32_t*>(pt + id_offset);
}
void launch_unpack_ring_scan(const uint8_t* raw, int64_t count, int stride, int xyz_offset, int stamp_offset, int id_offset, float4* xyz_out, double* stamp_out,
                             int32_t* id_out, hipStream_t s) {
    if (count > 0)
        hipLaunchKernelGGL(k_unpack_ring_scan, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, raw, count, stride, xyz_offset, stamp_offset, id_offset,
                           xyz_out, stamp_out, id_out);
}
void launch_pack_aos_window(const uint8_t* raw, int64_t count, int stride, int xyz_offset, int aux_offset, const int32_t* index, int row_limit, int fixed_row,
                            float4* local_out, int32_t* ring_out, int32_t* bad, hipStream_t s) {
    if (count > 0)
        hipLaunchKernelGGL(k_pack_aos_window, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, raw, count, stride, xyz_offset, aux_offset, index,
                           row_limit, fixed_row, local_out, ring_out, bad);
}
void launch_pack_aos_keyframe(const uint8_t* raw, int64_t count, int stride, int xyz_offset, int aux_offset, int row, float4* local_out, float4* normal_out,
                              hipStream_t s) {
    if (count > 0)
        hipLaunchKernelGGL(k_pack_aos_keyframe, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, raw, count, stride, xyz_offset, aux_offset, row,
                           local_out, normal_out);
}

}  // namespace dmsa
