// serial_kernels.hip — reference-order correspondence kernels for gfx950 (the library's default path).
//
// updateErrorTerms (DmsaOptimizer.h:234-273) per Gaussian k and evaluation b:
//     mean = 0; for j in members: mean += p_j            (float, member order, :247-254);  mean /= n
//     e    = 0; for j in members: e += double( float( (w d^T) A d ) ),  d = p_j - mean       (:259-264)
//     E[b][k] = sqrt(|e|)                                                                     (:267)
// with p_j = T_b[row_j] * local_j (the fused updateGlobalPoints, ContinuousTrajectory.h:151 / MapManagement.h:143).
// Both loops are serial chains whose rounding depends on the order, so the order is kept.  Mapping:
//   * lane = evaluation.  The members of a Gaussian are the same for every evaluation of a batch, so the lanes of a group walk
//     the member list together, each with its own pose table; a chain step is ONE vector add that serves every lane — no
//     cross-lane traffic, no reduction tree, full lane utilisation of the chain instructions.
//   * pose tables are stored transposed ([row][evaluation][12]) so that "row r of every evaluation" is one coalesced read, and
//     a lane keeps the 12 floats of its current row in registers: consecutive members of a Gaussian mostly share a row
//     (same scan, neighbouring firing times), so table traffic is a few percent of the member traffic.
//   * short Gaussians (k_residuals_small): a group of 16 / 32 / 64 lanes owns one Gaussian and does everything in registers.
//   * long Gaussians (k_residuals_chain): a dependent add has ~8 cycles latency and a single wave issues one instruction per
//     ~5 cycles, so 14 000 members x (transform + quadratic form + chain) in ONE wave would take milliseconds.  A workgroup
//     splits the roles instead: producer waves compute the per-member values of a 64-member chunk for all evaluations of the
//     sub-batch (lane = (member, evaluation)) into an LDS ring, and one chainer wave (lane = (coordinate, evaluation)) adds
//     them in member order, fed by ds_read_b128 (four members per read).  The critical path of a Gaussian is then its chain
//     alone: ~8 cycles per member for the float mean, ~12 for the double sum.
// Everything is compiled with -ffp-contract=off: separate multiplies and adds like the reference's -O1 build without FMA.
#include "serial_kernels.h"

#include "dev_sync.h"

#include <climits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace dmsa {

namespace {

constexpr int kSmallMax = 256;    // upper limit of the small-Gaussian threshold (histogram size of k_size_classes; = the 4 blocks of the fit's short class)

__device__ __forceinline__ float3 apply_row3s(const float4 r0, const float4 r1, const float4 r2, const float x, const float y, const float z) {
    float3 g;  // Matrix4f * Vector4f, column-wise like Eigen's packet product: ((c0*x + c1*y) + c2*z) + c3
    g.x = ((r0.x * x + r0.y * y) + r0.z * z) + r0.w;
    g.y = ((r1.x * x + r1.y * y) + r1.z * z) + r1.w;
    g.z = ((r2.x * x + r2.y * y) + r2.z * z) + r2.w;
    return g;
}
__device__ __forceinline__ float sum3s(float a, float b, float c) { return a + (b + c); }  // Eigen's 3-term redux
// the float Mahalanobis term of DmsaOptimizer.h:263: ((w d^T) A) d, every product and sum rounded separately
struct Info {
    float A00, A10, A20, A01, A11, A21, A02, A12, A22, w;
};
__device__ __forceinline__ Info load_info(const float4* __restrict__ info12, int g) {
    const float4 i0 = info12[3 * g], i1 = info12[3 * g + 1], i2 = info12[3 * g + 2];
    return Info{i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y};
}
__device__ __forceinline__ float mahalanobis_term(const Info& I, const float3 q, const float mx, const float my, const float mz) {
    const float d0 = q.x - mx, d1 = q.y - my, d2 = q.z - mz;
    const float wd0 = I.w * d0, wd1 = I.w * d1, wd2 = I.w * d2;
    const float v0 = sum3s(wd0 * I.A00, wd1 * I.A10, wd2 * I.A20);
    const float v1 = sum3s(wd0 * I.A01, wd1 * I.A11, wd2 * I.A21);
    const float v2 = sum3s(wd0 * I.A02, wd1 * I.A12, wd2 * I.A22);
    return sum3s(v0 * d0, v1 * d1, v2 * d2);
}
// LDS barrier that leaves global loads in flight (a __syncthreads() also waits for vmcnt(0), which would serialise the member
// prefetch of the producers with every phase)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Per-lane cache of one pose-table row (transposed tables: [row][B][12]).  The reload is written in assembly with its own
// s_waitcnt INSIDE the branch: left to the compiler, the wait for the three row loads lands behind the join of the branch as
// vmcnt(0), i.e. every step would also wait for the member prefetch that was issued just before it -- the full memory latency
// per step, taken or not.  This way only a step that really changes rows (a few percent) drains the queue.
// Packed fp32 (v_pk_mul_f32 / v_pk_add_f32 round each half separately, so it is legal here) is used in ONE place only, the quadratic
// form of the short tier.  Rounds 1-2 also packed the x / y halves of the transforms and the member pairs of the producers ("two
// thirds of the instructions"); on gfx950 a wave64 packed instruction occupies the SIMD for two passes, these kernels are bound by
// vector issue, and measured site by site in round 3 the transforms are faster written out as scalar operations
// (+1.7 % it/s), the parallel second pass is slower packed (-2.7 %), and the short tier's quadratic form is slower unpacked (-1.5 %).
typedef float f2 __attribute__((ext_vector_type(2)));
struct RowCache {
    int row = -1;
    f2 cx, cy, cz, cw;  // (row0[k], row1[k]) for k = x, y, z, w
    float4 r2;
    __device__ __forceinline__ void fetch(const float4* __restrict__ tabT, int B, int b, int want) {
        if (want != row) {
            const float4* t = tabT + ((size_t)want * B + b) * 3;
            float4 r0, r1;
            asm volatile(
                "global_load_dwordx4 %0, %3, off\n\t"
                "global_load_dwordx4 %1, %3, off offset:16\n\t"
                "global_load_dwordx4 %2, %3, off offset:32\n\t"
                "s_waitcnt vmcnt(0)"
                : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                : "v"(t)
                : "memory");
            cx = f2{r0.x, r1.x}, cy = f2{r0.y, r1.y}, cz = f2{r0.z, r1.z}, cw = f2{r0.w, r1.w};
            row = want;
        }
    }
    // Matrix4f * Vector4f, column-wise like Eigen's packet product: ((c0*x + c1*y) + c2*z) + c3
    __device__ __forceinline__ void apply(const float4 p, f2& gxy, float& gz) const {
        gxy.x = ((cx.x * p.x + cy.x * p.y) + cz.x * p.z) + cw.x;
        gxy.y = ((cx.y * p.x + cy.y * p.y) + cz.y * p.z) + cw.y;
        gz = ((r2.x * p.x + r2.y * p.y) + r2.z * p.z) + r2.w;
    }
};
// information matrix in the packed form of the short tier's quadratic form: columns 0 / 1 as pairs, column 2 and the weight as scalars
struct InfoPk {
    f2 a0, a1, a2;         // (A00, A01), (A10, A11), (A20, A21)
    float A02, A12, A22, w;
};
__device__ __forceinline__ InfoPk load_info_pk(const float4* __restrict__ info12, int g) {
    const float4 i0 = info12[3 * g], i1 = info12[3 * g + 1], i2 = info12[3 * g + 2];
    // info12 = A00 A10 A20 | A01 A11 A21 | A02 A12 A22 | w  (column-major 3x3 + weight)
    return InfoPk{f2{i0.x, i0.w}, f2{i0.y, i1.x}, f2{i0.z, i1.y}, i1.z, i1.w, i2.x, i2.y};
}
// the float Mahalanobis term of DmsaOptimizer.h:263, ((w d^T) A) d with 3-term sums as a + (b + c); identical operation order to
// mahalanobis_term(), the x / y columns packed
__device__ __forceinline__ float mahalanobis_term_pk(const InfoPk& I, const f2 gxy, const float gz, const f2 mxy, const float mz) {
    const f2 d = gxy - mxy;
    const float d2 = gz - mz;
    const f2 wd = I.w * d;
    const float wd2 = I.w * d2;
    const f2 v01 = wd.x * I.a0 + (wd.y * I.a1 + wd2 * I.a2);
    const float v2 = wd.x * I.A02 + (wd.y * I.A12 + wd2 * I.A22);
    const f2 t = v01 * d;
    return t.x + (t.y + v2 * d2);
}

// ---- member-pair form (producers of the chain kernel) -----------------------------------------------------------------
// A producer lane takes TWO consecutive members at a time (one LDS read of the member ring, one LDS write of the coordinates for
// both); when they share a pose-table row -- the common case: consecutive members of a Gaussian come from neighbouring firing times
// -- the row is loaded once.  The arithmetic is scalar per member (see RowCache).
typedef float f4 __attribute__((ext_vector_type(4)));
struct Rows {  // one pose-table row: three native 4-float vectors (inline-asm outputs stay in registers; a HIP float4 struct would not)
    f4 r0, r1, r2;
};
__device__ __forceinline__ Rows load_rows(const float4* __restrict__ tabT, int B, int b, int row) {
    const float4* t = tabT + ((size_t)row * B + b) * 3;
    Rows r;
    asm volatile(
        "global_load_dwordx4 %0, %3, off\n\t"
        "global_load_dwordx4 %1, %3, off offset:16\n\t"
        "global_load_dwordx4 %2, %3, off offset:32\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(r.r0), "=&v"(r.r1), "=&v"(r.r2)
        : "v"(t)
        : "memory");
    return r;
}
// Matrix4f * Vector4f, column-wise like Eigen's packet product: ((c0*x + c1*y) + c2*z) + c3; V = float or f2 (two members)
template <class V>
__device__ __forceinline__ void transform_v(const Rows rc, const V x, const V y, const V z, V& gx, V& gy, V& gz) {
    gx = ((rc.r0.x * x + rc.r0.y * y) + rc.r0.z * z) + rc.r0.w;
    gy = ((rc.r1.x * x + rc.r1.y * y) + rc.r1.z * z) + rc.r1.w;
    gz = ((rc.r2.x * x + rc.r2.y * y) + rc.r2.z * z) + rc.r2.w;
}
__device__ __forceinline__ void transform_v(const Rows rc, const f2 x, const f2 y, const f2 z, f2& gx, f2& gy, f2& gz) {  // a member pair
    float a, b, c, d, e, f;
    transform_v<float>(rc, x.x, y.x, z.x, a, b, c);
    transform_v<float>(rc, x.y, y.y, z.y, d, e, f);
    gx = f2{a, d}, gy = f2{b, e}, gz = f2{c, f};
}
// ((w d^T) A) d of DmsaOptimizer.h:263 with 3-term sums as a + (b + c) -- the operation order of mahalanobis_term()
template <class V>
__device__ __forceinline__ V mahalanobis_v(const Info& I, const V gx, const V gy, const V gz, const float mx, const float my, const float mz) {
    const V d0 = gx - mx, d1 = gy - my, d2 = gz - mz;
    const V wd0 = I.w * d0, wd1 = I.w * d1, wd2 = I.w * d2;
    const V v0 = wd0 * I.A00 + (wd1 * I.A10 + wd2 * I.A20);
    const V v1 = wd0 * I.A01 + (wd1 * I.A11 + wd2 * I.A21);
    const V v2 = wd0 * I.A02 + (wd1 * I.A12 + wd2 * I.A22);
    return v0 * d0 + (v1 * d1 + v2 * d2);
}
__device__ __forceinline__ f2 mahalanobis_v(const Info& I, const f2 gx, const f2 gy, const f2 gz, const float mx, const float my, const float mz) {
    return f2{mahalanobis_v<float>(I, gx.x, gy.x, gz.x, mx, my, mz), mahalanobis_v<float>(I, gx.y, gy.y, gz.y, mx, my, mz)};
}

}  // namespace

// ---- evaluations that need computing ------------------------------------------------------------------------------------------
// Evaluation b of a Jacobian batch differs from evaluation 0 in a contiguous range of pose-table rows, row_range[b] (k_eval_row_ranges:
// perturbing relative pose k of the keyframe chain leaves the frames in front of k untouched, ConsecutivePoses.h:26-43); a Gaussian
// reads the rows gauss_rows[g] (fit kernel).  If the two do not meet, every operand of the pair (g, b) has the bits of evaluation 0's
// and so has its residual: the pair is not computed (the Jacobian column kernel puts E[0][g] in its place).  The lanes of a group
// therefore do not stand for evaluations sub * L .. sub * L + L - 1 but for the sub-th run of L evaluations that DO need computing.
// Executed by the L lanes of a group (gshift: the group's first lane): list[r] = the evaluation of rank r0 + r among the active ones,
// r < cap; returns how many of the cap slots are filled.
template <int L>
__device__ __forceinline__ int build_eval_list(const int2* __restrict__ row_range, const int2 gr, int B, int r0, int cap, int* list, int bl, int gshift) {
    int base = 0;
    for (int c0 = 0; c0 < B; c0 += L) {
        const int b = c0 + bl;
        bool act = false;
        if (b < B) {
            const int2 e = row_range[b];
            act = b == 0 || !(e.y < gr.x || e.x > gr.y);
        }
        unsigned long long m = __ballot(act);
        if (L < 64) m = (m >> gshift) & ((1ull << (L & 63)) - 1ull);
        const int r = base + __popcll(m & ((1ull << bl) - 1ull)) - r0;
        if (act && r >= 0 && r < cap) list[r] = b;
        base += __popcll(m);
    }
    return min(max(base - r0, 0), cap);
}

// ------------------------------------------------------------------------------------------------------------
// size classes: one workgroup sorts the Gaussians by descending size class (counting sort in LDS)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_size_classes(const int32_t* __restrict__ seg_off, const GaussCounts* __restrict__ counts, int ns, int long_log2,
                                                       uint32_t* __restrict__ order, SerialCounts* __restrict__ out, const DevSync sy) {
    __shared__ int h_c[64], h_s[kSmallMax], s_max;
    dev_sync_enter(sy);  // e.g. the pose tables of the Jacobian batch, built beside the voxelisation: what follows this kernel on its stream needs them
    const int M = counts->level[0].num_gauss + counts->level[1].num_gauss;
    for (int i = threadIdx.x; i < 64; i += blockDim.x) h_c[i] = 0;
    for (int i = threadIdx.x; i < kSmallMax; i += blockDim.x) h_s[i] = 0;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    // chain bin: descending (floor(log2 n), next bit) -> at most 2x spread of sizes inside a bin, longest first
    auto chain_bin = [](int n) { const int k = 31 - __clz(n); const int sub = k > 0 ? (n >> (k - 1)) & 1 : 0; return 63 - (2 * k + sub); };
    // The sizes of the first kPer x 1024 Gaussians stay in registers for both passes, all their loads issued at once: the kernel is one
    // workgroup on the critical path, and a loop of dependent (load, LDS atomic) pairs pays the memory latency M / 1024 times.
    constexpr int kPer = 16;
    int nn[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int g = threadIdx.x + u * 1024;
        nn[u] = g < M ? max(seg_off[g + 1] - seg_off[g], 1) : 0;
    }
    int mx = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int n = nn[u];
        if (n > 0) {
            mx = max(mx, n);
            if (n <= ns)
                atomicAdd(&h_s[ns - n], 1);
            else
                atomicAdd(&h_c[chain_bin(n)], 1);
        }
    }
    for (int g = threadIdx.x + kPer * 1024; g < M; g += blockDim.x) {
        const int n = max(seg_off[g + 1] - seg_off[g], 1);
        mx = max(mx, n);
        if (n <= ns)
            atomicAdd(&h_s[ns - n], 1);
        else
            atomicAdd(&h_c[chain_bin(n)], 1);
    }
    atomicMax(&s_max, mx);
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive prefix over the 64 chain bins, then over the ns <= 256 short bins (kSmallMax / 64 per lane): one wave, two scans
        const int lane = threadIdx.x;
        auto scan = [&](int v) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(v, o);
                if (lane >= o) v += t;
            }
            return v;
        };
        const int c = h_c[lane];
        const int incl = scan(c);
        h_c[lane] = incl - c;
        const int n_chain = __shfl(incl, 63);
        const int n_long = __shfl(incl, 63 - 2 * long_log2);  // bins 0 .. 63 - 2 * long_log2 hold n >= 2^long_log2
        constexpr int kBinsPerLane = kSmallMax / 64;
        int v[kBinsPerLane], t = 0;
#pragma unroll
        for (int u = 0; u < kBinsPerLane; ++u) v[u] = kBinsPerLane * lane + u < ns ? h_s[kBinsPerLane * lane + u] : 0, t += v[u];
        const int ti = scan(t);
        int pos = n_chain + ti - t;
#pragma unroll
        for (int u = 0; u < kBinsPerLane; ++u) {
            if (kBinsPerLane * lane + u < ns) h_s[kBinsPerLane * lane + u] = pos;
            pos += v[u];
        }
        const int n_small = __shfl(ti, 63);
        if (lane == 0) out->n_chain = n_chain, out->n_small = n_small, out->max_members = s_max, out->n_long = n_long;
    }
    __syncthreads();
    // the order inside a bin is arbitrary: every (Gaussian, evaluation) result is independent of it
    int pos_u[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int n = nn[u];
        pos_u[u] = n <= 0 ? -1 : (n <= ns ? atomicAdd(&h_s[ns - n], 1) : atomicAdd(&h_c[chain_bin(n)], 1));
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u)
        if (pos_u[u] >= 0) order[pos_u[u]] = (uint32_t)(threadIdx.x + u * 1024);
    for (int g = threadIdx.x + kPer * 1024; g < M; g += blockDim.x) {
        const int n = max(seg_off[g + 1] - seg_off[g], 1);
        const int pos = n <= ns ? atomicAdd(&h_s[ns - n], 1) : atomicAdd(&h_c[chain_bin(n)], 1);
        order[pos] = (uint32_t)g;
    }
    dev_sync_leave(sy);  // the counts are final: their read-back (another stream) may start
}

__global__ __launch_bounds__(64) void k_eval_row_ranges(int model, const double* __restrict__ ctrl, int np, const double* __restrict__ stamps,
                                                        const double* __restrict__ traj_time, int n_t, int2* __restrict__ row_range) {
    __shared__ int s_rot[64];  // window: control rotation c of this evaluation differs from evaluation 0's
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b == 0) {
        if (lane == 0) row_range[0] = make_int2(0, INT_MAX);
        return;
    }
    const double* mine = ctrl + (size_t)b * np * 6;
    auto differs = [&](int pose, int first, int count) {
        bool d = false;
        for (int a = first; a < first + count; ++a) d = d || __double_as_longlong(mine[6 * pose + a]) != __double_as_longlong(ctrl[6 * pose + a]);
        return d;
    };
    int lo = INT_MAX, hi = -1;
    if (model == 2) {
        for (int k = lane; k < np; k += 64)
            if (differs(k, 0, 6)) lo = min(lo, k), hi = max(hi, k);
    } else {
        bool tr = false;
        for (int c = lane; c < np; c += 64) tr = tr || differs(c, 3, 3);
        const bool any_tr = __ballot(tr) != 0ull;
        if (any_tr) {
            lo = 0, hi = n_t - 1;  // every dense translation is a function of every control translation
        } else {
            for (int c = lane; c < min(np, 64); c += 64) s_rot[c] = differs(c, 0, 3) ? 1 : 0;
            __syncthreads();
            for (int j = lane; j < n_t; j += 64) {
                const double t = traj_time[j];
                int right = 0;  // getInterpRotation: lower_bound over stamps[0 .. C-2]
                while (right < np - 1 && stamps[right] < t) ++right;
                const bool d = right > 0 ? (s_rot[right - 1] | s_rot[right]) != 0 : s_rot[0] != 0;
                if (d) lo = min(lo, j), hi = max(hi, j);
            }
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lo = min(lo, __shfl_xor(lo, m)), hi = max(hi, __shfl_xor(hi, m));
    if (lane == 0) row_range[b] = make_int2(lo, hi);
}
void launch_eval_row_ranges(int model, const double* ctrl, int B, int np, const double* stamps, const double* traj_time, int n_t, int2* row_range, hipStream_t s) {
    if (B > 0) hipLaunchKernelGGL(k_eval_row_ranges, dim3(B), dim3(64), 0, s, model, ctrl, np, stamps, traj_time, n_t, row_range);
}

__global__ __launch_bounds__(256) void k_transpose_tables(const float4* __restrict__ tables, int rows, int B, float4* __restrict__ tablesT) {
    const int total = rows * B * 3;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x) {
        const int k = o % 3, rb = o / 3, b = rb % B, row = rb / B;
        tablesT[o] = tables[((size_t)b * rows + row) * 3 + k];
    }
}

// ------------------------------------------------------------------------------------------------------------
// short Gaussians: a group of L lanes = the evaluations of one (Gaussian, sub-batch); members are walked serially
// ------------------------------------------------------------------------------------------------------------
template <int L>
__device__ __forceinline__ void small_wave(const float4* __restrict__ memb, const int32_t* __restrict__ seg_off, const float4* __restrict__ info12,
                                           const float4* __restrict__ tabT, int B, const uint32_t* __restrict__ order, int n_items_g, int nsub,
                                           double* __restrict__ E, int64_t ldE, int wave_index, const int2* __restrict__ row_range,
                                           const int2* __restrict__ gauss_rows, int* list /* 64 ints of this wave */) {
    constexpr int G = 64 / L;  // Gaussians per wave (L = 9: seven, lane 63 idles)
    const int lane = threadIdx.x & 63, grp = lane / L, bl = lane % L;
    const int item = wave_index * G + grp;
    const int gi = grp < G ? item / nsub : n_items_g, sub = item - (item / nsub) * nsub;
    const int g = gi < n_items_g ? (int)order[gi] : 0;
    int b = sub * L + bl;
    bool on = gi < n_items_g && b < B;
    if (row_range != nullptr) {  // the sub-th run of L evaluations that can differ from evaluation 0 for this Gaussian (build_eval_list)
        const int2 gr = gi < n_items_g ? gauss_rows[g] : make_int2(INT_MAX, -1);
        const int cnt = build_eval_list<L>(row_range, gr, B, sub * L, L, list + grp * L, bl, grp * L);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        on = gi < n_items_g && bl < cnt;
        b = on ? list[grp * L + bl] : 0;
    }
    const int off0 = gi < n_items_g ? seg_off[g] : 0;
    const int n = on ? seg_off[g + 1] - off0 : 0;
    int nmax = n;  // longest member list of the wave's groups (sorted by size: nearly equal)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) nmax = max(nmax, __shfl_xor(nmax, m));
    if (nmax == 0) return;
    const int last = max(n - 1, 0);
    const int bc = on ? b : 0;
    RowCache rc;
    constexpr int U = 4;  // members in flight: every member register is refilled (for U steps ahead) right after its use, in place --
                          // rotating registers (cur = nxt) makes the compiler copy freshly loaded values, i.e. wait for them at once
    float4 m[U];
    auto member = [&](int jj) { return memb[off0 + min(jj, last)]; };
    // ---- float mean in member order (:247-254) ----
    float mx = 0.0f, my = 0.0f, mz = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) m[u] = member(u);
    for (int j0 = 0; j0 < nmax; j0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + u < n) {
                rc.fetch(tabT, B, bc, __float_as_int(m[u].w));
                f2 gxy;
                float gz;
                rc.apply(m[u], gxy, gz);
                mx = mx + gxy.x, my = my + gxy.y, mz = mz + gz;
            }
            m[u] = member(j0 + U + u);
        }
    }
    const float nf = (float)n;
    mx = mx / nf, my = my / nf, mz = mz / nf;
    // ---- float terms added to a double in member order (:259-264) ----
    const InfoPk I = load_info_pk(info12, g);
    const f2 mxy = f2{mx, my};
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) m[u] = member(u);
    for (int j0 = 0; j0 < nmax; j0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + u < n) {
                rc.fetch(tabT, B, bc, __float_as_int(m[u].w));
                f2 gxy;
                float gz;
                rc.apply(m[u], gxy, gz);
                acc += (double)mahalanobis_term_pk(I, gxy, gz, mxy, mz);
            }
            m[u] = member(j0 + U + u);
        }
    }
    if (on) E[(size_t)b * ldE + g] = sqrt(fabs(acc));
}

#ifndef DMSA_SMALL_WAVES
#define DMSA_SMALL_WAVES 1
#endif
template <int L>
__global__ __launch_bounds__(256, DMSA_SMALL_WAVES) void k_residuals_small(const float4* __restrict__ memb, const int32_t* __restrict__ seg_off,
                                                         const float4* __restrict__ info12, const float4* __restrict__ tabT, int B,
                                                         const uint32_t* __restrict__ order, int n_items_g, int nsub, double* __restrict__ E,
                                                         int64_t ldE, const int2* __restrict__ row_range, const int2* __restrict__ gauss_rows) {
    __shared__ int s_list[4][64];
    small_wave<L>(memb, seg_off, info12, tabT, B, order, n_items_g, nsub, E, ldE, (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)), row_range, gauss_rows,
                  s_list[threadIdx.x >> 6]);
}

// ------------------------------------------------------------------------------------------------------------
// long Gaussians: producers -> LDS ring -> chainer.  One workgroup = one (Gaussian, sub-batch of Bs evaluations).
//
// Phase p (one s_barrier per phase, kSlots = 3 ring slots): the producers fill slot p % 3 with chunk p while the chainer
// consumes chunk p - 2.  The lag of two phases is what lets the chainer run a rolling prefetch: chunk p - 1 is complete since
// the previous barrier, so the ds_reads for the next 16 members are always in flight while the current ones are added, across
// chunk boundaries too -- the chain never waits for LDS latency, only for its own dependent adds.
// ------------------------------------------------------------------------------------------------------------
#ifdef DMSA_SERIAL_TIMELINE
__device__ long long g_tl[2][16][64][2];
#define TL_BARRIER(pass, ph)                                                             \
    do {                                                                                 \
        const int _ph = (ph);                                                            \
        if (kSepLoader && blockIdx.x == 0 && _ph >= 32 && _ph < 96) {                                  \
            const long long _a = clock64();                                              \
            lds_barrier();                                                               \
            const long long _b = clock64();                                              \
            if ((threadIdx.x & 63) == 0) g_tl[pass][threadIdx.x >> 6][_ph - 32][0] = _a, g_tl[pass][threadIdx.x >> 6][_ph - 32][1] = _b; \
        } else                                                                           \
            lds_barrier();                                                               \
    } while (0)
#else
#define TL_BARRIER(pass, ph) lds_barrier()
#endif
constexpr int kSlots = 3;

// ---- second pass without a chain ---------------------------------------------------------------------------------------------
// The reference adds the float terms of a Gaussian to a double one by one (DmsaOptimizer.h:259-264): acc = RN(acc + t_j), acc_0 = 0.
// Let all terms be > 0 and q = min_j (exponent(t_j) - 23): every term -- a float, 24 significant bits -- is a multiple of 2^q, and
// so is every partial sum, in member order or in ANY other order; none exceeds U = sum_j t_j.  If U < 2^(q+53) all those partial
// sums are representable doubles, no addition rounds, and the chain, a tree and the exact sum are the same number.  That is the
// case for practically every Gaussian (rebalanced sums stay below 2^7, the smallest terms are ~2^-18: 49 of 53 bits), so the
// second pass is computed as a plain parallel reduction that also tracks the smallest term, and the condition is CHECKED with
// integer arithmetic on conservative bounds (U <= computed sum x (1 + 2^-30)).  A (Gaussian, sub-batch) that fails -- a zero or
// negative term, a tiny term beside a large sum -- is summed again one member after the other by the pipelined code below.
__device__ unsigned long long g_fallback_sums;  // (Gaussian, sub-batch) workgroups that failed the test (tree_mode 1) since the last reset
__device__ __forceinline__ uint32_t hi_word(double x) { return (uint32_t)(__double_as_longlong(x) >> 32); }

// members per phase of the latency tier: 128 (two LDS-DMA instructions per chunk, 82 KB of LDS, one workgroup per CU -- there are fewer
// latency-tier workgroups than CUs) halves the barriers per member of the 64 the ring started with: 11 -> 9.8 cycles per member
#ifndef DMSA_LONG_CHUNK
#define DMSA_LONG_CHUNK 128
#endif
constexpr int kBL = 16;  // evaluation stride of the LDS ring layout (compile time: every ds_read of the chainer gets an immediate offset)
// what k_residuals_chain needs to share the second pass of the longest Gaussians out to helper workgroups (host side: LongSplit)
struct LongHelp {
    float* means = nullptr;      // [item][3][kBL], written by the item's chain workgroup
    uint32_t* ready = nullptr;   // [item]: == epoch once the means of this launch are there
    double* partial = nullptr;   // [item][helper][kBL]
    int* partial_key = nullptr;  // [item][helper][kBL]
    uint32_t* done = nullptr;    // [item]: tickets of the helpers, back to zero when the item is finished
    int32_t* timed_out = nullptr;
    uint32_t epoch = 0;
    int min_members = 0, helpers = 0, items = 0;  // items: the chain workgroups of the launch (the helpers follow them in the grid)
    int lead = 0;                                 // only the first `lead` Gaussians of the order have helpers
};
__device__ __forceinline__ void dev_sync_wait_eq(const uint32_t* word, uint32_t want, int32_t* timed_out) {
    uint32_t v = 0;
    for (int spin = 0; spin < kSyncMaxSpins; ++spin) {
        v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == want) return;
        if (spin < 4096)
            __builtin_amdgcn_s_sleep(2);
        else
            __builtin_amdgcn_s_sleep(16);
    }
    if (timed_out != nullptr) timed_out[0] = 1, timed_out[1] = (int32_t)want, timed_out[2] = (int32_t)v;
}

// kProd producer waves; kSepLoader: one more wave that only feeds the member ring (otherwise the last producer does that too).
//   <8, true, 128> latency tier (the longest Gaussians): with Bs <= 8 every producer has ONE step per phase, so a phase lasts as
//              long as its chain segment even for the producer wave that loses the issue arbitration on the chainer's SIMD
//   <4, false, 32> throughput tier: 5 waves and 20 KB of LDS per workgroup (32-member chunks), six workgroups per CU; phases are
//              producer-bound, but the resident waves spend most of their time issuing instead of waiting at a barrier
template <int kProd, bool kSepLoader, int kChunk>
// the throughput tier runs at 8 waves per SIMD (64 VGPRs, a few spilled dwords): +3 % per iteration over the 6 waves the compiler picks on its own
#ifndef DMSA_MID_WAVES
#define DMSA_MID_WAVES 8
#endif
#ifndef DMSA_LONG_WAVES
#define DMSA_LONG_WAVES 1
#endif
__global__ __launch_bounds__(64 * (kProd + 1 + (kSepLoader ? 1 : 0)), kSepLoader ? DMSA_LONG_WAVES : DMSA_MID_WAVES) void k_residuals_chain(
    const float4* __restrict__ memb, const int32_t* __restrict__ seg_off, const float4* __restrict__ info12, const float4* __restrict__ tabT, int B,
    const uint32_t* __restrict__ order, int Bs, int nsub, int prio, int tree_mode, double* __restrict__ E, int64_t ldE, uint32_t* start_signal,
    const uint32_t* __restrict__ rot_same, const int2* __restrict__ row_range, const int2* __restrict__ gauss_rows,
    const LongHelp help /* means != null: the longest Gaussians' second pass is shared out to HELPER workgroups at the end of this launch's grid */) {
    // tree_mode 0: second pass as a chain (the reference's loop, pipelined); 1: parallel second pass, chain only if the exactness test
    // fails; 2: parallel pass computed, then the chain anyway (test hook)
    constexpr int kProducers = kProd;
    constexpr int kWaves = kProd + 1 + (kSepLoader ? 1 : 0);
    constexpr int kMaxSteps = (kChunk / 8 + kProd - 1) / kProd;  // steps per chunk <= kChunk / 8 (Bs = kBL: 4 member pairs per step)
    // pass 1 ring: float  q[kSlots][kChunk / 4][3][kBL][4]   (member-in-group fastest: the chainer reads four members per ds_read_b128)
    // pass 2 ring: double t[kSlots][kChunk / 2][kBL][2]       (aliases the pass-1 ring)
    constexpr int kSlotFloats = kChunk * 3 * kBL, kSlotDoubles = kChunk * kBL;
    __shared__ __attribute__((aligned(16))) float s_q[kSlots * kSlotFloats];
    __shared__ float s_mean[3 * kBL];
    __shared__ __attribute__((aligned(16))) float4 s_m[4][kChunk];  // member ring (filled by LDS-DMA)
    __shared__ int s_chain;  // the parallel second pass failed its test for some evaluation
    __shared__ int s_last;   // helper: this workgroup took the item's last ticket
    // latency tier: the sums and smallest terms of the BLOCKS of the member list (a wave's share of the parallel second pass, or a helper's
    // slice), kept for the block-wise fall-back below when the test on the whole Gaussian fails
    constexpr int kBlocks = kSepLoader ? (kWaves > 8 ? kWaves : 8) : 1;
    __shared__ double s_bsum[kBlocks][kBL];
    __shared__ int s_bkey[kBlocks][kBL];
    __shared__ int s_evals[kBL], s_nb;  // the evaluations of this workgroup's lanes (build_eval_list) and how many there are
    double* s_t = reinterpret_cast<double*>(s_q);

    // Fork of the tier streams (launch_sync_wait in front of the other tiers): the LAST workgroup of the latency tier releases them, i.e.
    // they start once every workgroup of this launch has its CU -- released any earlier, thousands of their workgroups get in first and
    // the longest Gaussians, which bound the batch, queue behind them (+60 us per line-search batch).
    // Helpers.  The longest Gaussian bounds a batch twice over: its float chain (pass 1: ~8 cycles x 14 000 members on ONE wave, 57 us that no
    // arrangement shortens) and, behind it, its second pass on the same ten waves (40 us for 5 evaluations, 165 us for 16) while its neighbours have
    // long finished.  With `help`, the workgroup of a Gaussian of at least help.min_members members ENDS with its chain: it leaves the means in
    // device memory and raises a flag; help.helpers more workgroups per such item -- the blocks behind the help.items chain workgroups of this
    // launch, dispatched after them -- wait for the flag and sum a slice of the members each (the parallel second pass below: same lanes, same
    // exactness argument -- when the integer bounds hold no addition rounds, so ANY order is the reference's chain).  The last helper to arrive
    // adds the slices in a fixed order, tests the bounds and writes E, or runs the member-by-member chain itself if the test fails.
    const bool helping = kSepLoader && help.means != nullptr;  // (the latency tier only: compile-time dead in the throughput tier)
    const bool helper = helping && (int)blockIdx.x >= help.items;
    if (start_signal != nullptr && (int)blockIdx.x == (helping ? help.items : (int)gridDim.x) - 1 && threadIdx.x == 0)
        __hip_atomic_fetch_add(start_signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int item = blockIdx.x, hslot = 0;
    if (helper) {
        const int hb = (int)blockIdx.x - help.items;
        item = hb / help.helpers, hslot = hb - item * help.helpers;
    }
    const int gi = item / nsub, sub = item - gi * nsub;
    const int g = (int)order[gi];
    const int off0 = seg_off[g], n = seg_off[g + 1] - off0;
    const bool hand_over = helping && n >= help.min_members && gi < help.lead;  // this item's second pass belongs to its helpers
    if (helper && !hand_over) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave == 0) {
        int cnt;
        if (row_range != nullptr) {  // the sub-th run of Bs evaluations that can differ from evaluation 0 for this Gaussian
            cnt = build_eval_list<64>(row_range, gauss_rows[g], B, sub * Bs, Bs, s_evals, lane, 0);
        } else {
            cnt = min(Bs, B - sub * Bs);
            if (lane < Bs) s_evals[lane] = sub * Bs + lane;
        }
        if (lane == 0) s_nb = cnt;
    }
    __syncthreads();
    const int nb = s_nb;
    if (nb <= 0) return;  // every evaluation of this sub-batch equals evaluation 0 for this Gaussian (or the Gaussian has fewer sub-batches)
    const int nchunks = (n + kChunk - 1) / kChunk;
    const int nphases = nchunks + 2;

    // ---- second pass as a parallel reduction (all waves; see "second pass without a chain" above) ----
    // lane = (member of the step, evaluation) like the producers; every wave strides through the member list on its own, partial sums
    // and the smallest term stay in registers, one LDS reduction per workgroup at the end.  Returns true when E is written.
    auto parallel_second_pass = [&]() -> bool {
        if (tree_mode == 0) return false;
        const int mpl = 64 / Bs, ms2 = lane / Bs, pb2 = lane - ms2 * Bs;  // members per wave step
        const bool lane_on2 = ms2 < mpl && pb2 < nb;
        const int bcol2 = s_evals[pb2 < nb ? pb2 : 0];
        const float mx2 = s_mean[pb2 & 15], my2 = s_mean[kBL + (pb2 & 15)], mz2 = s_mean[2 * kBL + (pb2 & 15)];
        const Info I2 = load_info(info12, g);
        if (threadIdx.x == 0) s_chain = 0;
        double part = 0.0;
        int key = 0x7fffffff;  // bits of the smallest term (positive floats order like their bit patterns); <= 0: a zero or negative term (or -NaN)
        Rows r2;
        r2.r0 = r2.r1 = r2.r2 = f4{0.0f, 0.0f, 0.0f, 0.0f};
        int r2_row = -1;
        // the members this workgroup sums: all of them, or the helper's slice; a contiguous block of that range per wave
        int rb = 0, re = n;
        if (helper) {
            const int blk_h = ((n + help.helpers - 1) / help.helpers + mpl - 1) / mpl * mpl;
            rb = min(n, hslot * blk_h), re = min(n, rb + blk_h);
        }
        const int blk = ((re - rb + kWaves - 1) / kWaves + mpl - 1) / mpl * mpl;
        const int wbeg0 = rb + wave * blk;
        const int jend = min(re, wbeg0 + blk);
        // ---- sub-batch of evaluations that share their ROTATIONS with evaluation 0 (forward differences of translation parameters: half
        // of the Jacobian batch).  ((c0 x + c1 y) + c2 z) + c3 has the same first three terms for all of them: S = (c0 x + c1 y) + c2 z is
        // computed ONCE per member, lane = member, from evaluation 0's rows, parked in a wave-private kilobyte of the idle ring, and a
        // step is S + c3 of the lane's own evaluation and the quadratic form: 38 instead of 53 vector instructions per member step for
        // the same operations on the same operands in the same order.  The flags come from the pose-table kernels, which compare the
        // control rotations bit for bit -- nothing is assumed about the model (the window with IMU rows never qualifies: its round trip
        // through the preintegration factors touches relative pose 0 in every evaluation).
        bool shared_rot = false;
        if (rot_same != nullptr) shared_rot = __ballot(lane < nb && rot_same[s_evals[lane]] == 0u) == 0ull;
        if (shared_rot) {
            static_assert((size_t)kWaves * (48 + 64) * 16 <= sizeof(float) * kSlots * kSlotFloats, "second-pass staging does not fit the ring");
            float4* stage = reinterpret_cast<float4*>(s_q) + kWaves * 48 + wave * 64;  // behind the reduction scratch (kWaves x 64 x 12 bytes)
            const int CH = (64 / mpl) * mpl, steps = CH / mpl;  // members per chunk: a multiple of the members per step
            const int wbeg = wbeg0;
            float tx = 0.0f, ty = 0.0f, tz = 0.0f;  // translation column of the lane's evaluation at row t_row
            int t_row = -1;
            auto fetch_member = [&](int cs) { return memb[off0 + min(cs + lane, n - 1)]; };
            float4 pre = fetch_member(wbeg);
            for (int cs = wbeg; cs < jend; cs += CH) {
                const float4 m = pre;
                pre = fetch_member(cs + CH);
                {
                    const float4* t0 = tabT + (size_t)__float_as_int(m.w) * B * 3;  // evaluation 0 of the batch
                    const float4 q0 = t0[0], q1 = t0[1], q2 = t0[2];
                    stage[lane] = float4{(q0.x * m.x + q0.y * m.y) + q0.z * m.z, (q1.x * m.x + q1.y * m.y) + q1.z * m.z, (q2.x * m.x + q2.y * m.y) + q2.z * m.z, m.w};
                }
                __builtin_amdgcn_wave_barrier();  // (compiler only) one wave's LDS instructions execute in order; the lanes read each other's members
                for (int st = 0; st < steps; ++st) {
                    const int jl = st * mpl + ms2, jj = cs + jl;
                    if (lane_on2 && jj < jend) {
                        const float4 v = stage[jl];
                        const int row = __float_as_int(v.w);
                        if (row != t_row) {
                            const float* tr = reinterpret_cast<const float*>(tabT + ((size_t)row * B + bcol2) * 3);
                            tx = tr[3], ty = tr[7], tz = tr[11], t_row = row;
                        }
                        const float t = mahalanobis_v(I2, v.x + tx, v.y + ty, v.z + tz, mx2, my2, mz2);
                        part += (double)t;
                        key = min(key, __float_as_int(t));
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // every wave takes one contiguous block of the member list (consecutive members mostly share a pose-table row, so a lane
        // changes rows about once per row of its block instead of once per step); members are fetched two steps ahead
        int j = wbeg0 + ms2;
        auto term = [&](const float4 m, int jj) {
            if (lane_on2 && jj < jend) {
                const int row = __float_as_int(m.w);
                if (row != r2_row) r2 = load_rows(tabT, B, bcol2, row), r2_row = row;
                float gx, gy, gz;
                transform_v(r2, m.x, m.y, m.z, gx, gy, gz);
                const float t = mahalanobis_v(I2, gx, gy, gz, mx2, my2, mz2);
                const double td = (double)t;
                part += td;
                key = min(key, __float_as_int(t));
            }
        };
        // four member registers used in turn, each refilled (for four steps ahead) right AFTER its use: the load can land in the
        // register it replaces, so there is no rotation copy -- a copy of a freshly loaded register makes the step wait for its own load
        auto fetch = [&](int jj) { return memb[off0 + min(jj, n - 1)]; };
        float4 m_a = fetch(j), m_b = fetch(j + mpl), m_c = fetch(j + 2 * mpl), m_d = fetch(j + 3 * mpl);
        for (int j0 = shared_rot ? jend : wbeg0; j0 < jend; j0 += 4 * mpl) {
            term(m_a, j), m_a = fetch(j + 4 * mpl);
            term(m_b, j + mpl), m_b = fetch(j + 5 * mpl);
            term(m_c, j + 2 * mpl), m_c = fetch(j + 6 * mpl);
            term(m_d, j + 3 * mpl), m_d = fetch(j + 7 * mpl);
            j += 4 * mpl;
        }
        double* red = s_t;  // the rings are idle between the passes
        int* redk = reinterpret_cast<int*>(s_t + kWaves * 64);
        red[wave * 64 + lane] = part, redk[wave * 64 + lane] = key;
        lds_barrier();
        if (helper) {
            // the slice goes to device memory; the workgroup that takes the item's last ticket adds all slices (fixed order) and goes on
            if (wave == 0 && lane < nb) {
                double U = 0.0;
                int k = 0x7fffffff;
                for (int w = 0; w < kWaves; ++w)
                    for (int s2 = 0; s2 < mpl; ++s2) U += red[w * 64 + s2 * Bs + lane], k = min(k, redk[w * 64 + s2 * Bs + lane]);
                const size_t at = ((size_t)item * help.helpers + hslot) * kBL + lane;
                __hip_atomic_store(help.partial + at, U, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(help.partial_key + at, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wave == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                if (lane == 0) {
                    const uint32_t tk = __hip_atomic_fetch_add(help.done + item, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                    s_last = tk == (uint32_t)help.helpers - 1u ? 1 : 0;
                }
            }
            lds_barrier();
            if (s_last == 0) return true;  // somebody else finishes the item
            if (wave == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (lane == 0) __hip_atomic_store(help.done + item, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
                if (lane < nb) {
                    for (int hh = 0; hh < help.helpers; ++hh) {
                        const size_t at = ((size_t)item * help.helpers + hh) * kBL + lane;
                        red[hh * 64 + lane] = __hip_atomic_load(help.partial + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        redk[hh * 64 + lane] = __hip_atomic_load(help.partial_key + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        if (wave == 0 && lane < nb) {
            double U = 0.0;
            int k = 0x7fffffff;
            if (helper) {
                for (int hh = 0; hh < help.helpers; ++hh) {
                    U += red[hh * 64 + lane], k = min(k, redk[hh * 64 + lane]);
                    if constexpr (kSepLoader)
                        if (hh < kBlocks) s_bsum[hh][lane] = red[hh * 64 + lane], s_bkey[hh][lane] = redk[hh * 64 + lane];
                }
            } else {
                for (int w = 0; w < kWaves; ++w) {
                    double ub = 0.0;
                    int kb = 0x7fffffff;
                    for (int s2 = 0; s2 < mpl; ++s2) ub += red[w * 64 + s2 * Bs + lane], kb = min(kb, redk[w * 64 + s2 * Bs + lane]);
                    U += ub, k = min(k, kb);
                    if constexpr (kSepLoader) s_bsum[w][lane] = ub, s_bkey[w][lane] = kb;
                }
            }
            const int q = ((k >> 23) & 0xff) - 127 - 23;        // every term is a multiple of 2^q (a denormal smallest term: of 2^-149, so also of 2^-150)
            const int pe = min(max(q + 53 + 1023, 0), 2046);
            const double limit = __hiloint2double(pe << 20, 0);  // 2^(q+53); 0 when that is below the normal range: the test fails
            // k <= 0: some term was zero or negative -- the proof needs positive terms, so the chain runs (a non-positive U would otherwise
            // slip under the 2^-993 the clamped exponent gives).  A NaN term makes U a NaN, which fails the comparison; U itself may be
            // rounded: < n 2^-53 relative
            const bool exact = k > 0 && U * (1.0 + 0x1p-30) < limit && tree_mode == 1;
            if (exact)
                E[(size_t)s_evals[lane] * ldE + g] = sqrt(fabs(U));
            else
                s_chain = 1;
        }
        lds_barrier();
        const bool chain = s_chain != 0;
        if (chain && threadIdx.x == 0 && tree_mode == 1) atomicAdd(&g_fallback_sums, 1ull);
        return !chain;
    };

    if (wave == 0) {
        // ================= chainer: lane = (coordinate, evaluation) =================
        // A single wave issues one instruction per ~5 cycles, so the chain loop carries nothing but the adds and one ds_read per
        // four (two) members: no exec masking (lanes outside the (coordinate, evaluation) grid add garbage that is never
        // stored), immediate LDS offsets, fully unrolled chunks.
        __builtin_amdgcn_s_setprio(3);  // the chain is the critical path of the workgroup (and of the launch for the longest Gaussians)
        const int cc = lane >> 4, cb = lane & 15;
        const bool on1 = cc < 3 && cb < nb;
        float acc = 0.0f;
        if (helper) {
            // wait for the chain workgroup of this item (it was dispatched before this block), then take its means
            if (lane == 0) dev_sync_wait_eq(help.ready + item, help.epoch, help.timed_out);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (lane < 3 * kBL) s_mean[lane] = __hip_atomic_load(help.means + (size_t)item * (3 * kBL) + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const float4* q4 = reinterpret_cast<const float4*>(s_q) + (cc < 3 ? cc * kBL + cb : 0);
            constexpr int gstride = 3 * kBL;       // float4 entries per group of four members
            constexpr int slot4 = kSlotFloats / 4;
#ifndef DMSA_CHAIN_DEPTH_SHORT
#define DMSA_CHAIN_DEPTH_SHORT 8
#endif
            // prefetch depth and steps per chunk (one step = one float4 = four members)
            constexpr int S = kChunk / 4, D = kChunk >= 64 ? 8 : DMSA_CHAIN_DEPTH_SHORT;
            static_assert(S % D == 0, "the register ring r[k % D] runs on across chunk boundaries: D must divide the steps of a chunk");
            float4 r[D];
            lds_barrier();  // the member ring holds chunk 0
            lds_barrier();  // phase 0: the producers fill chunk 0
            for (int p = 1; p < nphases; ++p) {
                const int c = p - 2;
                if (p == 1) {
#pragma unroll
                    for (int k = 0; k < D; ++k) r[k] = q4[k * gstride];  // chunk 0 is complete since the first barrier
                } else {
                    const float4* cs = q4 + (c % kSlots) * slot4;
                    const float4* ns = q4 + ((c + 1) % kSlots) * slot4;  // complete since the previous barrier (or unused)
                    const int cnt = min(kChunk, n - c * kChunk);
                    if (cnt == kChunk) {
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const float4 v = r[k % D];
                            r[k % D] = k + D < S ? cs[(k + D) * gstride] : ns[(k + D - S) * gstride];
                            acc = acc + v.x, acc = acc + v.y, acc = acc + v.z, acc = acc + v.w;
                            __builtin_amdgcn_sched_barrier(0);  // keep ONE read per four adds, D steps ahead (the scheduler would cluster the reads and wait)
                        }
                    } else {  // last chunk of the Gaussian
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const float4 v = r[k % D];
                            if (k + D < S) r[k % D] = cs[(k + D) * gstride];
                            if (4 * k + 0 < cnt) acc = acc + v.x;
                            if (4 * k + 1 < cnt) acc = acc + v.y;
                            if (4 * k + 2 < cnt) acc = acc + v.z;
                            if (4 * k + 3 < cnt) acc = acc + v.w;
                        }
                    }
                }
                TL_BARRIER(0, p);
            }
        }
        if (!helper && on1) s_mean[cc * kBL + cb] = acc / (float)n;
        lds_barrier();
        if (!helper && hand_over) {  // the second pass is the helpers' work: publish the means, raise the flag, leave
            if (on1) __hip_atomic_store(help.means + (size_t)item * (3 * kBL) + cc * kBL + cb, acc / (float)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(help.ready + item, help.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (parallel_second_pass()) return;
        const bool on2 = lane < nb;  // coordinate slot 0
        // the chained second pass over `sn` members (a segment of the list, or all of it), continuing the sums `dacc`: the phases the
        // producers run in produce_pass2 below, barrier for barrier
        auto chain_pass2 = [&](const int sn, double dacc, const bool primed) __attribute__((always_inline)) -> double {
            const int sphases = (sn + kChunk - 1) / kChunk + 2;
            const double2* t2 = reinterpret_cast<const double2*>(s_t) + cb;
            constexpr int slot2 = kSlotDoubles / 2;
            // (depth 4 for the 32-member chunks of the throughput tier: with 8 the sixteen double2 registers of this ring pushed 23 VGPRs of the
            // 64-register budget into scratch -- in this loop only, which runs when the exactness test of the parallel second pass fails)
#ifndef DMSA_CHAIN2_DEPTH_SHORT
#define DMSA_CHAIN2_DEPTH_SHORT 4
#endif
            constexpr int S = kChunk / 2, D = kChunk >= 64 ? 8 : DMSA_CHAIN2_DEPTH_SHORT;  // one step = one double2 = two members
            static_assert(S % D == 0, "the register ring r[k % D] runs on across chunk boundaries: D must divide the steps of a chunk");
            double2 r[D];
            if (primed) lds_barrier();  // (pairs with the producers' "the member ring holds the segment's chunk 0")
            lds_barrier();  // phase 0
            for (int p = 1; p < sphases; ++p) {
                const int c = p - 2;
                if (p == 1) {
#pragma unroll
                    for (int k = 0; k < D; ++k) r[k] = t2[k * kBL];
                } else {
                    const double2* cs = t2 + (c % kSlots) * slot2;
                    const double2* ns = t2 + ((c + 1) % kSlots) * slot2;
                    const int cnt = min(kChunk, sn - c * kChunk);
                    if (cnt == kChunk) {
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const double2 v = r[k % D];
                            r[k % D] = k + D < S ? cs[(k + D) * kBL] : ns[(k + D - S) * kBL];
                            dacc += v.x, dacc += v.y;
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < S; ++k) {
                            const double2 v = r[k % D];
                            if (k + D < S) r[k % D] = cs[(k + D) * kBL];
                            if (2 * k + 0 < cnt) dacc += v.x;
                            if (2 * k + 1 < cnt) dacc += v.y;
                        }
                    }
                }
                TL_BARRIER(1, p);
            }
            return dacc;
        };
        if constexpr (kSepLoader) {
            if ((tree_mode == 1 || tree_mode == 3) && (!helper || help.helpers <= kBlocks)) {
                // ---- block-wise fall-back (latency tier).  The test on the whole Gaussian failed: some term is too small beside the sum (or not
                // positive).  The parallel pass left the sum s_b and the smallest term of every BLOCK of the member list (a wave's share, a
                // helper's slice), blocks in member order.  Walking the blocks with the running sum S of the reference's chain: if S is a
                // multiple of 2^a, every term of the block a multiple of 2^q_b (positive floats), g = min(a, q_b) and S + s_b < 2^(g+53), then
                // every partial sum of the chain through this block is a multiple of 2^g below 2^(g+53) -- a representable double: no addition
                // rounds, the chain leaves the block with exactly S + s_b (and s_b itself is exact by the same bound).  A block that fails --
                // the one with the tiny term; a block in which S, carrying low bits from an earlier rounding, crosses a power of two -- is
                // chained member by member from S on.  One ill-placed member of a 14 000-member Gaussian then costs a tenth of its chain, not
                // all of it (measured on the bench window: a fixed point with such a member ran 7 % slower per iteration).
                // tree_mode 3 (test hook): every block through the chain.
                const int mplb = 64 / Bs, nblk = helper ? help.helpers : kWaves;
                const int bsz = ((n + nblk - 1) / nblk + mplb - 1) / mplb * mplb;
                lds_barrier();  // every wave has read the verdict on the whole Gaussian (s_chain) before the blocks' verdicts overwrite it
                double S = 0.0;
                for (int w = 0; w < nblk; ++w) {
                    const int beg = min(n, w * bsz), cntm = min(n, beg + bsz) - beg;
                    if (cntm <= 0) continue;
                    double sb = 0.0;
                    bool ok = true;
                    if (on2) {
                        sb = s_bsum[w][lane];
                        const int key = s_bkey[w][lane];
                        ok = key > 0;
                        int gq = ((key >> 23) & 0xff) - 127 - 23;  // every term of the block is a multiple of 2^gq ...
                        if (S != 0.0) {                             // ... and so is S, once gq is lowered to its lowest set bit
                            const long long bits = __double_as_longlong(S);
                            const int e = (int)((bits >> 52) & 0x7ff);
                            const unsigned long long m = ((unsigned long long)bits & 0xfffffffffffffull) | (1ull << 52);
                            gq = min(gq, e - 1023 - 52 + (int)__builtin_ctzll(m));
                            ok = ok && e != 0 && e != 0x7ff && bits > 0;  // a denormal, infinite, NaN or negative S: the chain decides
                        }
                        const int pe = min(max(gq + 53 + 1023, 0), 2046);
                        const double limit = __hiloint2double(pe << 20, 0);  // 2^(gq+53)
                        ok = ok && (S + sb) * (1.0 + 0x1p-30) < limit;       // (sb may be rounded if the block is not exact: < n 2^-53 relative; NaN fails)
                    }
                    const bool bad = tree_mode == 3 || __ballot(on2 && !ok) != 0ull;
                    if (lane == 0) s_chain = bad ? 1 : 0;
                    lds_barrier();  // A: the producers read the verdict
                    if (bad)
                        S = chain_pass2(cntm, S, true);
                    else
                        S = S + sb;
                    lds_barrier();  // B: ... before the next block's overwrites it
                }
                if (on2) E[(size_t)s_evals[lane] * ldE + g] = sqrt(fabs(S));
                return;
            }
        }
        if (helper) lds_barrier();  // (pairs with the producers' "the member ring holds chunk 0 again")
        const double dacc = chain_pass2(n, 0.0, false);
        if (on2) E[(size_t)s_evals[lane] * ldE + g] = sqrt(fabs(dacc));
        return;
    }

    // ================= producers: lane = (member of the step, evaluation) =================
    // the workgroups of the longest Gaussians bound the launch: their producers win the VALU arbitration on their CU
    if (prio >= 2)
        __builtin_amdgcn_s_setprio(2);
    else if (prio == 1)
        __builtin_amdgcn_s_setprio(1);
    const int pw = wave - 1;
    const bool loader = pw == (kSepLoader ? kProducers : kProducers - 1);  // the wave that feeds the member ring
    const bool works = !kSepLoader || pw < kProducers;
    const int mps = 64 / Bs;                              // member PAIRS per step
    const int steps = (kChunk / 2 + mps - 1) / mps;       // steps per chunk
    const int ms = lane / Bs, pb = lane - ms * Bs;
    const bool lane_on = ms < mps && pb < nb;
    const int bcol = s_evals[pb < nb ? pb : 0];
    const int last = n - 1;
    Rows rc;  // the pose-table row of this lane's evaluation that the last member used, and its index
    int rc_row = -1;
    rc.r0 = rc.r1 = rc.r2 = f4{0.0f, 0.0f, 0.0f, 0.0f};
    int jl_u[kMaxSteps];  // first member (even) of this lane's pair in step u (kChunk: none)
#pragma unroll
    for (int u = 0; u < kMaxSteps; ++u) {
        const int t = pw + u * kProducers, jl = 2 * (t * mps + ms);
        jl_u[u] = (works && t < steps && lane_on && jl < kChunk) ? jl : kChunk;
    }
    // Members reach the producers through LDS: ONE wave issues ONE global_load_lds_dwordx4 per chunk (64 lanes x 16 B = the 64
    // members of a chunk, written lane-linear into a 4-slot ring) two phases ahead of its use.  No member ever sits in a
    // register across a barrier, so there is nothing for the compiler to copy or to wait for inside the phase loop; the only
    // vector-memory wait of a phase is the loader's "all but the newest DMA have landed" in front of the barrier.  Slots past the
    // end of the Gaussian hold copies of its last member (clamped addresses): whatever a lane computes from them is never read.
    const unsigned lds_m = (unsigned)(uintptr_t)&s_m[0][0];
    auto dma = [&](int P) {  // members of the chunk consumed in global phase P (pass 1: P = p, pass 2: P = nphases + p)
        if (loader) {
            const int c = P < nphases ? P : P - nphases;
#pragma unroll
            for (int hh = 0; hh < kChunk / 64 + (kChunk % 64 ? 1 : 0); ++hh) {  // 64 members (lanes) per LDS-DMA instruction
                if (hh * 64 + lane < kChunk) {
                    const float4* src = memb + off0 + min(c * kChunk + hh * 64 + lane, last);
                    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_m + (unsigned)(P & 3) * (kChunk * 16) + (unsigned)hh * 1024u);
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep)
                                 : "v"(src), "s"(dst)
                                 : "memory");
                }
            }
        }
    };
    constexpr int kDmaPerChunk = kChunk / 64 + (kChunk % 64 ? 1 : 0);
    auto landed = [&]() {  // every DMA but the newest chunk's has landed (one chunk is issued per phase, always)
        if (loader) {
            if constexpr (kDmaPerChunk == 1)
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
    };
    // the pair (jl, jl + 1) of ring slot `slot`, coordinates as member pairs: ds_read2_b32 with dword offsets (k, k + 4)
    struct Pair {
        f2 x, y, z;
        int row0, row1;
    };
    auto read_pair = [&](int slot, int jl) {
        const float* m = reinterpret_cast<const float*>(&s_m[slot][jl]);
        return Pair{f2{m[0], m[4]}, f2{m[1], m[5]}, f2{m[2], m[6]}, __float_as_int(m[3]), __float_as_int(m[7])};
    };
    // ---- pass 1: transformed coordinates of every member ----
    if (!helper) {
    dma(0), dma(1);
    landed();
    lds_barrier();  // chunk 0 of the member ring is visible
    for (int p = 0; p < nphases; ++p) {
        dma(p + 2);
        if (p < nchunks) {
            float* slot = s_q + (p % kSlots) * kSlotFloats + pb * 4;
            const int left = n - p * kChunk;  // members from this chunk on
#pragma unroll
            for (int u = 0; u < kMaxSteps; ++u) {
                if (jl_u[u] < left && jl_u[u] < kChunk) {
                    const int jl = jl_u[u];
                    const Pair m = read_pair(p & 3, jl);
                    float* dst = slot + (jl >> 2) * (3 * kBL * 4) + (jl & 3);  // jl even: the pair is two adjacent floats
                    if (m.row0 != rc_row) rc = load_rows(tabT, B, bcol, m.row0), rc_row = m.row0;
                    f2 gx, gy, gz;
                    transform_v(rc, m.x, m.y, m.z, gx, gy, gz);
                    if (m.row1 != m.row0) {  // a row boundary between the two members: the second one again, with its own row
                        const Rows r1 = load_rows(tabT, B, bcol, m.row1);
                        float hx, hy, hz;
                        transform_v(r1, m.x.y, m.y.y, m.z.y, hx, hy, hz);
                        gx.y = hx, gy.y = hy, gz.y = hz;
                    }
                    *reinterpret_cast<f2*>(dst) = gx, *reinterpret_cast<f2*>(dst + 4 * kBL) = gy, *reinterpret_cast<f2*>(dst + 8 * kBL) = gz;
                }
            }
        }
        landed();
        TL_BARRIER(0, p);
    }
    }  // (!helper)
    // ---- pass 2: Mahalanobis terms (needs the mean of pass 1) ----
    const Info I = load_info(info12, g);
    lds_barrier();  // s_mean is complete
    if (!helper && hand_over) {
        if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the member DMAs issued ahead
        return;
    }
    if (parallel_second_pass()) {
        if (loader && !helper) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the member DMAs issued ahead for a chained second pass
        return;
    }
    // The chained second pass is the rare path (a failed exactness test).  What it needs of the lane -- its member slot, its evaluation, its
    // steps -- is derived AGAIN here from an opaque copy of the lane index, and the pose-table row cache starts empty: nothing of the producers'
    // per-lane state stays live across the parallel second pass above.  (At 64 registers the compiler otherwise parks six VGPRs in scratch
    // memory around it, in every producer wave of every workgroup: 10 MB of stores per launch of the throughput tier in round 5.)
    int lane2;
    asm volatile("v_and_b32 %0, 63, %1" : "=v"(lane2) : "v"(threadIdx.x));
    const int ms_2 = lane2 / Bs, pb_2 = lane2 - ms_2 * Bs;
    const bool lane_on_2 = ms_2 < mps && pb_2 < nb;
    const int bcol_2 = s_evals[pb_2 < nb ? pb_2 : 0];
    Rows rc2;
    rc2.r0 = rc2.r1 = rc2.r2 = f4{0.0f, 0.0f, 0.0f, 0.0f};
    int rc2_row = -1;
    const float mx = s_mean[pb_2 & 15], my = s_mean[kBL + (pb_2 & 15)], mz = s_mean[2 * kBL + (pb_2 & 15)];
    // the terms of the members [sbeg, sbeg + sn) of the Gaussian into the ring, phase by phase (chain_pass2 of the chainer consumes them).
    // pbase: the global phase the segment's first chunk was (or is) loaded for -- the whole list after pass 1: nphases, its first two chunks
    // are in flight since the last phases of pass 1; a segment of the block-wise fall-back (prime): 0, the ring is drained and loaded afresh
    auto produce_pass2 = [&](const int sbeg, const int sn, const int pbase, const bool prime) __attribute__((always_inline)) {
        const int schunks = (sn + kChunk - 1) / kChunk, sphases = schunks + 2, slast = sbeg + sn - 1;
        auto sdma = [&](int P, int c) {  // chunk c of the segment for global phase P
            if (loader) {
#pragma unroll
                for (int hh = 0; hh < kChunk / 64 + (kChunk % 64 ? 1 : 0); ++hh) {
                    if (hh * 64 + lane < kChunk) {
                        const float4* src = memb + off0 + min(sbeg + c * kChunk + hh * 64 + lane, slast);
                        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_m + (unsigned)(P & 3) * (kChunk * 16) + (unsigned)hh * 1024u);
                        unsigned keep;
                        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                     : "=&s"(keep)
                                     : "v"(src), "s"(dst)
                                     : "memory");
                    }
                }
            }
        };
        if (prime) {
            if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // whatever was loaded ahead for another range
            sdma(pbase, 0), sdma(pbase + 1, 1);
            landed();
            lds_barrier();  // the member ring holds the segment's chunk 0
        }
        for (int p = 0; p < sphases; ++p) {
            const int P = pbase + p;
            sdma(P + 2, p + 2);
            if (p < schunks) {
                double* slot = s_t + (p % kSlots) * kSlotDoubles + pb_2 * 2;
                const int left = sn - p * kChunk;
#pragma unroll
                for (int u = 0; u < kMaxSteps; ++u) {
                    const int t_2 = pw + u * kProducers, jl_2 = 2 * (t_2 * mps + ms_2);
                    const int jl = (works && t_2 < steps && lane_on_2 && jl_2 < kChunk) ? jl_2 : kChunk;  // (jl_u[u] of pass 1)
                    if (jl < left && jl < kChunk) {
                        const Pair m = read_pair(P & 3, jl);
                        double* dst = slot + (jl >> 1) * (kBL * 2);  // jl even: the pair is one double2
                        if (m.row0 != rc2_row) rc2 = load_rows(tabT, B, bcol_2, m.row0), rc2_row = m.row0;
                        f2 gx, gy, gz;
                        transform_v(rc2, m.x, m.y, m.z, gx, gy, gz);
                        if (m.row1 != m.row0) {
                            const Rows r1 = load_rows(tabT, B, bcol_2, m.row1);
                            float hx, hy, hz;
                            transform_v(r1, m.x.y, m.y.y, m.z.y, hx, hy, hz);
                            gx.y = hx, gy.y = hy, gz.y = hz;
                        }
                        const f2 t = mahalanobis_v(I, gx, gy, gz, mx, my, mz);
                        *reinterpret_cast<double2*>(dst) = double2{(double)t.x, (double)t.y};
                    }
                }
            }
            landed();
            TL_BARRIER(1, p);
        }
    };
    if constexpr (kSepLoader) {
        if ((tree_mode == 1 || tree_mode == 3) && (!helper || help.helpers <= kBlocks)) {  // the block-wise fall-back (see the chainer)
            const int mplb = 64 / Bs, nblk = helper ? help.helpers : kWaves;
            const int bsz = ((n + nblk - 1) / nblk + mplb - 1) / mplb * mplb;
            lds_barrier();  // (the chainer's "every wave has read the verdict on the whole Gaussian")
            for (int w = 0; w < nblk; ++w) {
                const int beg = min(n, w * bsz), cntm = min(n, beg + bsz) - beg;
                if (cntm <= 0) continue;
                lds_barrier();  // A
                const bool bad = s_chain != 0;
                if (bad) produce_pass2(beg, cntm, 0, true);
                lds_barrier();  // B
            }
            if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land in LDS after the workgroup has released it
            return;
        }
    }
    if (helper) {  // the chain after all: what pass 1 would have left in flight -- the first two chunks of the member ring
        dma(nphases), dma(nphases + 1);
        landed();
        lds_barrier();
    }
    produce_pass2(0, n, nphases, false);
    if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may land in LDS after the workgroup has released it
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
// members; Gaussians up to this size go to the lane-per-evaluation kernel, which spends fewer instructions per member than the chain
// tiers (no ring, no barriers) as long as its waves stay short: measured 96 / 128 / 192 / 256 / 384 / 512 -> 1242 / 1256 / 1267 / 1272 /
// 1241 / 1201 it/s on the bench window (128 until the end of round 3)
int serial_small_threshold() { return 256; }
#ifndef DMSA_LONG_LOG2
#define DMSA_LONG_LOG2 12
#endif
static int serial_long_log2() { return DMSA_LONG_LOG2; }  // Gaussians with >= 2^12 members go to the latency tier
unsigned long long serial_fallback_sums(bool reset) {
    unsigned long long v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_fallback_sums), sizeof(v));
    if (reset) {
        const unsigned long long z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fallback_sums), &z, sizeof(z));
    }
    return v;
}
SerialShape serial_shape(int B) {
    // evaluations per workgroup of the two chain tiers.  The 9 trials of the line search go through the latency tier as 5 + 4: its
    // longest Gaussian bounds that batch, and half the evaluations per workgroup halve its (workgroup-local) second pass: +1.5 % it/s.
    // (3 + 3 + 3 and 2 x 5 for the throughput tier, or 4 x 8 for the Jacobian batch, are slower: the first pass is repeated per sub-batch.)
    constexpr int kTrialLong = 5;
    const int bs_long = B <= kBL ? kTrialLong : kBL, bs_mid = kBL;
    SerialShape s;
    s.nsub_long = (B + bs_long - 1) / bs_long, s.Bs_long = (B + s.nsub_long - 1) / s.nsub_long;
    s.nsub = (B + bs_mid - 1) / bs_mid, s.Bs = (B + s.nsub - 1) / s.nsub;
#ifndef DMSA_SMALL_LANES_9
#define DMSA_SMALL_LANES_9 0
#endif
    // (experiment, off: the nine trials of the line search as seven Gaussians per wave with nine lanes each -- 63 of 64 lanes busy instead of
    // four groups with nine of sixteen -- measured 1 % SLOWER per iteration: the waves carry seven member lists instead of four and end later)
    s.lanes = (DMSA_SMALL_LANES_9 && B == 9) ? 9 : (B <= 16 ? 16 : (B <= 32 ? 32 : 64));
    s.nsub_small = (B + s.lanes - 1) / s.lanes;
    return s;
}
void launch_size_classes(const int32_t* seg_off, const GaussCounts* counts, uint32_t* order, SerialCounts* out, hipStream_t s, const DevSync& sy, int small_threshold,
                         int long_log2) {
    const int ns = small_threshold > 0 ? (small_threshold < kSmallMax ? small_threshold : kSmallMax) : serial_small_threshold();
    const int ll = long_log2 >= 9 && long_log2 <= 20 ? long_log2 : serial_long_log2();
    hipLaunchKernelGGL(k_size_classes, dim3(1), dim3(1024), 0, s, seg_off, counts, ns, ll, order, out, sy);
}
void launch_transpose_tables(const float* tables, int rows, int B, float* tablesT, hipStream_t s) {
    const int total = rows * B * 3;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_transpose_tables, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0, s, reinterpret_cast<const float4*>(tables), rows, B,
                       reinterpret_cast<float4*>(tablesT));
}
void launch_residuals_serial(const float4* memb_local, const int32_t* seg_off, const float* info12, const float* tablesT, int B, const uint32_t* order,
                             const SerialCounts& sc, double* E, int64_t ldE, hipStream_t s_long, hipStream_t s_rest, hipStream_t s_small, int tree_mode,
                             uint32_t* start_signal, int tiers, const uint32_t* rot_same, const int2* row_range, const int2* gauss_rows, const LongSplit* split) {
    if (B <= 0) return;
    const SerialShape sh = serial_shape(B);
    const float4* info = reinterpret_cast<const float4*>(info12);
    const float4* tabT = reinterpret_cast<const float4*>(tablesT);
    const int n_long = sc.n_long, n_mid = sc.n_chain - sc.n_long;
    // latency tier first (its longest chain bounds the batch), blocks in descending size; the other tiers fill the chip around it
    if (n_long > 0 && (tiers & 1)) {
        const unsigned items = (unsigned)n_long * sh.nsub_long;
        LongHelp help{};
        unsigned grid = items;
        if (split != nullptr && split->means != nullptr && tree_mode != 0) {
            help.means = split->means, help.ready = split->ready, help.partial = split->partial, help.partial_key = split->partial_key, help.done = split->done;
            help.timed_out = split->timed_out;
            help.epoch = split->epoch, help.min_members = split->min_members, help.helpers = split->helpers, help.items = (int)items;
            // helpers for the leading Gaussians of the order (descending size): the ones that may reach min_members
            help.lead = std::min(n_long, split->lead_gaussians);
            grid += (unsigned)help.lead * sh.nsub_long * (unsigned)split->helpers;
        }
        hipLaunchKernelGGL((k_residuals_chain<8, true, DMSA_LONG_CHUNK>), dim3(grid), dim3(64 * 10), 0, s_long, memb_local, seg_off, info, tabT, B, order,
                           sh.Bs_long, sh.nsub_long, 2, tree_mode, E, ldE, start_signal, rot_same, row_range, gauss_rows, help);
    }
    if (n_mid > 0 && (tiers & 2))
        hipLaunchKernelGGL((k_residuals_chain<4, false, 32>), dim3((unsigned)n_mid * sh.nsub), dim3(64 * 5), 0, s_rest, memb_local, seg_off, info, tabT, B,
                           order + n_long, sh.Bs, sh.nsub, 0, tree_mode, E, ldE, (uint32_t*)nullptr, rot_same, row_range, gauss_rows, LongHelp{});
    if (sc.n_small > 0 && (tiers & 4)) {
        const int items = sc.n_small * sh.nsub_small;
        const int per_block = 4 * (64 / sh.lanes);
        const dim3 grid((items + per_block - 1) / per_block);
        const uint32_t* ord = order + sc.n_chain;
        if (sh.lanes == 9)
            hipLaunchKernelGGL(k_residuals_small<9>, grid, dim3(256), 0, s_small, memb_local, seg_off, info, tabT, B, ord, sc.n_small, sh.nsub_small, E, ldE, row_range, gauss_rows);
        else if (sh.lanes == 16)
            hipLaunchKernelGGL(k_residuals_small<16>, grid, dim3(256), 0, s_small, memb_local, seg_off, info, tabT, B, ord, sc.n_small, sh.nsub_small, E, ldE, row_range, gauss_rows);
        else if (sh.lanes == 32)
            hipLaunchKernelGGL(k_residuals_small<32>, grid, dim3(256), 0, s_small, memb_local, seg_off, info, tabT, B, ord, sc.n_small, sh.nsub_small, E, ldE, row_range, gauss_rows);
        else
            hipLaunchKernelGGL(k_residuals_small<64>, grid, dim3(256), 0, s_small, memb_local, seg_off, info, tabT, B, ord, sc.n_small, sh.nsub_small, E, ldE, row_range, gauss_rows);
    }
#ifdef DMSA_SERIAL_TIMELINE
    if (n_long > 0) {
        static long long h[2][16][64][2];
        (void)hipStreamSynchronize(s_long);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tl), sizeof(h));
        for (int pass = 0; pass < 2; ++pass) {
            fprintf(stderr, "[timeline] B=%d pass=%d chainer work per phase (release -> next arrival), phases 40..55:", B, pass + 1);
            for (int ph = 8; ph < 24; ++ph) fprintf(stderr, " %lld", h[pass][0][ph + 1][0] - h[pass][0][ph][1]);
            fprintf(stderr, "  | phase length (arrival -> arrival):");
            for (int ph = 8; ph < 24; ++ph) fprintf(stderr, " %lld", h[pass][0][ph + 1][0] - h[pass][0][ph][0]);
            fprintf(stderr, "\n");
        }
    }
#endif
}

}  // namespace dmsa
