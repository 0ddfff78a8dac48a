// dmsa_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the DMSA hot path.
//
//   K0 transform           ContinuousTrajectory::updateGlobalPoints :137-155 / MapManagement::updateGlobalPoints :140-147
//   K1 pose tables         ContinuousTrajectory::updateTrajDenseTforms :193-225 (control chain stays on the host)
//   K2 voxel lattice/keys  pcl::octree::OctreePointCloud as used by DmsaOptimizer::createGaussianSets :282-298
//   K3 Gaussian fit        Gaussians::addPointSet / limitCovariance / updateRebalancingWeights (Gaussians.h:130-201)
//   K4 correspondence      DmsaOptimizer::updateErrorTerms :242-268 fused with the transform, B pose tables per launch
//   K5 normal equations    DmsaOptimizer.h:107-113 (H = J^T J, g = J^T e) + e^T e for the line search :171
//
// Everything here is HBM/LDS-bound integer and fp32/fp64 vector work; none of it is a contraction worth MFMA at
// P = 30.  Built with -ffp-contract=off (no FMA fusion: the transform and the voxel keys
// must round exactly like the reference's separate multiply/add).
#include "dmsa_kernels.h"
#include "radix_sort_dev.h"

#include <cfloat>
#include <climits>
#include <type_traits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cstdio>

#include "../../include/dmsa_detmath.h"
#include "../../include/dmsa_hip.h"

namespace dmsa {

// ------------------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------------------
// DPP (data-parallel primitive) lane movement keeps wave-wide sums in the VALU instead of round trips through the LDS
// crossbar (ds_bpermute): row_shr:1/2/4/8 inside each 16-lane row, then row_bcast:15 / row_bcast:31 across rows (gfx9).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kCtrl, kRowMask, 0xf, true));
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ int dpp_mov(int v) {
    return __builtin_amdgcn_update_dpp(0, v, kCtrl, kRowMask, 0xf, true);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_mov(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <class T>
__device__ __forceinline__ T wave_incl_scan_dpp(T v) {
    v += dpp_mov<0x111, 0xf>(v);  // row_shr:1
    v += dpp_mov<0x112, 0xf>(v);  // row_shr:2
    v += dpp_mov<0x114, 0xf>(v);  // row_shr:4
    v += dpp_mov<0x118, 0xf>(v);  // row_shr:8
    v += dpp_mov<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v += dpp_mov<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return v;
}
__device__ __forceinline__ float wave_allsum(float v) {
    v = wave_incl_scan_dpp(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double wave_allsum(double v) {
    v = wave_incl_scan_dpp(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ int wave_allmin(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_allminf(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_allmaxf(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// Matrix4f * Vector4f with w == 1, evaluated column-wise like Eigen's packet product: ((c0*x + c1*y) + c2*z) + c3
__device__ __forceinline__ float3 apply_row3(const float4 r0, const float4 r1, const float4 r2, const float x, const float y, const float z) {
    float3 g;
    g.x = ((r0.x * x + r0.y * y) + r0.z * z) + r0.w;
    g.y = ((r1.x * x + r1.y * y) + r1.z * z) + r1.w;
    g.z = ((r2.x * x + r2.y * y) + r2.z * z) + r2.w;
    return g;
}
__device__ __forceinline__ float sum3f(float a, float b, float c) { return a + (b + c); }
// tile slots of a Gaussian with n members: rounded up to the 8 slots a thread of the tiled kernels owns (see k_tile_rows)
__host__ __device__ __forceinline__ int pad_slots(int n) { return (n + 7) & ~7; }

// ------------------------------------------------------------------------------------------------------------
// K0 — transforms
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transform(const float4* __restrict__ local, const float4* __restrict__ table, float4* __restrict__ global,
                                                   int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 p = local[i];
        const int row = __float_as_int(p.w);
        const float4 r0 = table[3 * row], r1 = table[3 * row + 1], r2 = table[3 * row + 2];
        const float3 g = apply_row3(r0, r1, r2, p.x, p.y, p.z);
        global[i] = make_float4(g.x, g.y, g.z, 1.0f);
    }
}
__global__ __launch_bounds__(256) void k_transform_normals(const float4* __restrict__ local, const float4* __restrict__ nlocal,
                                                           const float4* __restrict__ table, float4* __restrict__ global, float4* __restrict__ nglobal,
                                                           int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 p = local[i];
        const float4 v = nlocal[i];
        const int row = __float_as_int(p.w);
        const float4 r0 = table[3 * row], r1 = table[3 * row + 1], r2 = table[3 * row + 2];
        const float3 g = apply_row3(r0, r1, r2, p.x, p.y, p.z);
        global[i] = make_float4(g.x, g.y, g.z, 1.0f);
        // Matrix3f * Vector3f: coefficient-wise 3-term inner products, x0 + (x1 + x2)
        nglobal[i] = make_float4(sum3f(r0.x * v.x, r0.y * v.y, r0.z * v.z), sum3f(r1.x * v.x, r1.y * v.y, r1.z * v.z),
                                 sum3f(r2.x * v.x, r2.y * v.y, r2.z * v.z), 0.0f);
    }
}
__global__ __launch_bounds__(256) void k_shift_points(float4* __restrict__ pts, int64_t n, float ox, float oy, float oz, float sign) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 p = pts[i];
    if (sign > 0.0f)
        p.x = p.x + ox, p.y = p.y + oy, p.z = p.z + oz;
    else
        p.x = p.x - ox, p.y = p.y - oy, p.z = p.z - oz;
    pts[i] = p;
}

static inline int grid_for(int64_t n, int block, int cap = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

void launch_transform(const float4* local, const float4* table, float4* global, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_transform, dim3(grid_for(n, 256)), dim3(256), 0, s, local, table, global, n);
}
void launch_transform_normals(const float4* local, const float4* nlocal, const float4* table, float4* global, float4* nglobal, int64_t n,
                              hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_transform_normals, dim3(grid_for(n, 256)), dim3(256), 0, s, local, nlocal, table, global, nglobal, n);
}
void launch_shift_points(float4* pts, int64_t n, float ox, float oy, float oz, float sign, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_shift_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, pts, n, ox, oy, oz, sign);
}

// ------------------------------------------------------------------------------------------------------------
// K1 — dense pose tables (double math, one thread per dense pose).  sin / cos / acos / atan2 come from include/dmsa_detmath.h:
// fixed sequences of correctly rounded IEEE operations, so the tables are bit-identical to the host's and the oracle's.
// ------------------------------------------------------------------------------------------------------------
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ void d_so3_exp(const D3 w, double R[9]) {
    const double theta = sqrt(w.x * w.x + w.y * w.y + w.z * w.z);
    if (theta < 0.00001) {
        R[0] = 1, R[1] = 0, R[2] = 0, R[3] = 0, R[4] = 1, R[5] = 0, R[6] = 0, R[7] = 0, R[8] = 1;
        return;
    }
    const double s = dmsa_det::det_sin(theta) / theta;
    const double sh = dmsa_det::det_sin(0.5 * theta);
    const double c = 2.0 * sh * sh / (theta * theta);
    const double t2 = theta * theta;
    R[0] = 1.0 + c * (w.x * w.x - t2);
    R[4] = 1.0 + c * (w.y * w.y - t2);
    R[8] = 1.0 + c * (w.z * w.z - t2);
    R[1] = c * w.x * w.y - s * w.z;
    R[3] = c * w.x * w.y + s * w.z;
    R[2] = c * w.x * w.z + s * w.y;
    R[6] = c * w.x * w.z - s * w.y;
    R[5] = c * w.y * w.z - s * w.x;
    R[7] = c * w.y * w.z + s * w.x;
}
__device__ __forceinline__ void d_quat_from_axang(const D3 a, double q[4]) {
    const double sq = a.x * a.x + a.y * a.y + a.z * a.z;
    const double ang = sqrt(sq);
    D3 ax = a;
    if (sq > 0.0) ax = D3{a.x / ang, a.y / ang, a.z / ang};
    const double sh = dmsa_det::det_sin(0.5 * ang);
    q[0] = dmsa_det::det_cos(0.5 * ang), q[1] = sh * ax.x, q[2] = sh * ax.y, q[3] = sh * ax.z;
}
// slerp of two rotations given as the unit quaternions d_quat_from_axang makes of them (helpers.h:24-37)
__device__ __forceinline__ D3 d_slerp_quat(const double* q1, const double* q2, const double t) {
    const double one = 1.0 - DBL_EPSILON;
    const double d = q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2] + q1[3] * q2[3];
    const double ad = fabs(d);
    double s0, s1;
    if (ad >= one) {
        s0 = 1.0 - t, s1 = t;
    } else {
        const double th = dmsa_det::det_acos(ad), sn = dmsa_det::det_sin(th);
        s0 = dmsa_det::det_sin((1.0 - t) * th) / sn;
        s1 = dmsa_det::det_sin(t * th) / sn;
    }
    if (d < 0.0) s1 = -s1;
    const double qw = s0 * q1[0] + s1 * q2[0], qx = s0 * q1[1] + s1 * q2[1], qy = s0 * q1[2] + s1 * q2[2], qz = s0 * q1[3] + s1 * q2[3];
    double n = sqrt(qx * qx + qy * qy + qz * qz);
    if (n == 0.0) return D3{0.0, 0.0, 0.0};
    const double angle = 2.0 * dmsa_det::det_atan2(n, fabs(qw));
    if (qw < 0.0) n = -n;
    return D3{(qx / n) * angle, (qy / n) * angle, (qz / n) * angle};
}

constexpr int kMaxCtrl = 64;  // control poses per window the table kernel keeps in LDS

__global__ __launch_bounds__(256) void k_window_pose_tables(const double* __restrict__ ctrl, const double* __restrict__ stamps,
                                                            const double* __restrict__ fh_w, const double* __restrict__ traj_time, int C, int n_t,
                                                            float* __restrict__ tables, float* __restrict__ tablesT, uint32_t* __restrict__ rot_same) {
    __shared__ double s_ctrl[kMaxCtrl * 6];
    __shared__ double s_stamp[kMaxCtrl];
    __shared__ double s_w[kMaxCtrl];
    __shared__ double s_quat[kMaxCtrl * 4];  // Quaterniond(AngleAxisd) of every control pose: the same value for every dense pose that uses it
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * 6; i += blockDim.x) s_ctrl[i] = ctrl[(size_t)b * C * 6 + i];
    for (int i = threadIdx.x; i < C; i += blockDim.x) s_stamp[i] = stamps[i], s_w[i] = fh_w[i];
    if (rot_same != nullptr && blockIdx.x == 0) {
        // rot_same[b] = 1: every control ROTATION of evaluation b has the bits of evaluation 0's (a forward difference of a translation
        // parameter), so every dense rotation of its table has them too -- same inputs, same instructions (serial_kernels.hip shares the
        // rotated coordinates of a member among such evaluations)
        bool diff = false;
        for (int i = threadIdx.x; i < C * 3; i += blockDim.x) {
            const int c = i / 3, a = i - 3 * c;
            diff = diff || __double_as_longlong(ctrl[(size_t)b * C * 6 + 6 * c + a]) != __double_as_longlong(ctrl[6 * c + a]);
        }
        const int any = __syncthreads_or(diff ? 1 : 0);
        if (threadIdx.x == 0) rot_same[b] = any ? 0u : 1u;
    } else {
        __syncthreads();
    }
    if (threadIdx.x < C) {
        const double* a = &s_ctrl[6 * threadIdx.x];
        d_quat_from_axang(D3{a[0], a[1], a[2]}, &s_quat[4 * threadIdx.x]);
    }
    __syncthreads();
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_t) return;
    float* out = tables + ((size_t)b * (n_t + 1) + j) * 12;
    // the reference-order correspondence kernels read the tables as [row][evaluation][12]: written here as well, when asked for
    float* outT = tablesT ? tablesT + ((size_t)j * gridDim.y + b) * 12 : nullptr;
    if (j == n_t) {  // identity row for static points
        out[0] = 1, out[1] = 0, out[2] = 0, out[3] = 0, out[4] = 0, out[5] = 1, out[6] = 0, out[7] = 0, out[8] = 0, out[9] = 0, out[10] = 1, out[11] = 0;
        if (outT)
            for (int q = 0; q < 12; ++q) outT[q] = out[q];
        return;
    }
    const double t = traj_time[j];
    // getInterpRotation: lower_bound over stamps[0 .. C-2]
    int right = 0;
    while (right < C - 1 && s_stamp[right] < t) ++right;
    D3 o;
    if (right > 0) {
        const double t_rel = (t - s_stamp[right - 1]) / (s_stamp[right] - s_stamp[right - 1]);
        o = d_slerp_quat(&s_quat[4 * (right - 1)], &s_quat[4 * right], t_rel);
    } else {
        o = D3{s_ctrl[0], s_ctrl[1], s_ctrl[2]};
    }
    // Floater–Hormann evaluation with the exact-node short-circuit, one interpolant per axis (shared weights)
    // (the weight quotient w_i / (t - x_i) and the denominator are the same numbers for the three axes: computed once)
    double tr[3];
    {
        double num[3] = {0.0, 0.0, 0.0}, den = 0.0, exact[3] = {0.0, 0.0, 0.0};
        bool hit = false;
        for (int i = 0; i < C; ++i) {
            if (t == s_stamp[i]) {
                if (!hit)
                    for (int a = 0; a < 3; ++a) exact[a] = s_ctrl[6 * i + 3 + a];
                hit = true;
            }
            if (!hit) {
                const double q = s_w[i] / (t - s_stamp[i]);
                for (int a = 0; a < 3; ++a) num[a] += q * s_ctrl[6 * i + 3 + a];
                den += q;
            }
        }
        for (int a = 0; a < 3; ++a) tr[a] = hit ? exact[a] : num[a] / den;
    }
    double R[9];
    d_so3_exp(o, R);
    out[0] = (float)R[0], out[1] = (float)R[1], out[2] = (float)R[2], out[3] = (float)tr[0];
    out[4] = (float)R[3], out[5] = (float)R[4], out[6] = (float)R[5], out[7] = (float)tr[1];
    out[8] = (float)R[6], out[9] = (float)R[7], out[10] = (float)R[8], out[11] = (float)tr[2];
    if (outT) {
        float4* o4 = reinterpret_cast<float4*>(outT);
        o4[0] = make_float4(out[0], out[1], out[2], out[3]), o4[1] = make_float4(out[4], out[5], out[6], out[7]), o4[2] = make_float4(out[8], out[9], out[10], out[11]);
    }
}

__global__ __launch_bounds__(256) void k_keyframe_pose_tables(const double* __restrict__ frames, int F, float* __restrict__ tables, float* __restrict__ tablesT,
                                                              uint32_t* __restrict__ rot_same) {
    const int b = blockIdx.y;
    if (rot_same != nullptr && blockIdx.x == 0) {  // as in k_window_pose_tables: the frame rotations of evaluation b against evaluation 0's
        bool diff = false;
        for (int i = threadIdx.x; i < F * 3; i += blockDim.x) {
            const int c = i / 3, a = i - 3 * c;
            diff = diff || __double_as_longlong(frames[((size_t)b * F + c) * 6 + a]) != __double_as_longlong(frames[(size_t)c * 6 + a]);
        }
        const int any = __syncthreads_or(diff ? 1 : 0);
        if (threadIdx.x == 0) rot_same[b] = any ? 0u : 1u;
    }
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > F) return;
    float* out = tables + ((size_t)b * (F + 1) + k) * 12;
    float* outT = tablesT ? tablesT + ((size_t)k * gridDim.y + b) * 12 : nullptr;
    if (k == F) {
        out[0] = 1, out[1] = 0, out[2] = 0, out[3] = 0, out[4] = 0, out[5] = 1, out[6] = 0, out[7] = 0, out[8] = 0, out[9] = 0, out[10] = 1, out[11] = 0;
        if (outT)
            for (int q = 0; q < 12; ++q) outT[q] = out[q];
        return;
    }
    const double* p = frames + ((size_t)b * F + k) * 6;
    double R[9];
    d_so3_exp(D3{p[0], p[1], p[2]}, R);
    out[0] = (float)R[0], out[1] = (float)R[1], out[2] = (float)R[2], out[3] = (float)p[3];
    out[4] = (float)R[3], out[5] = (float)R[4], out[6] = (float)R[5], out[7] = (float)p[4];
    out[8] = (float)R[6], out[9] = (float)R[7], out[10] = (float)R[8], out[11] = (float)p[5];
    if (outT)
        for (int q = 0; q < 12; ++q) outT[q] = out[q];
}

__global__ __launch_bounds__(256) void k_detmath_eval(int fn, const double* __restrict__ x, const double* __restrict__ y, int64_t n, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i];
    out[i] = fn == 0 ? dmsa_det::det_sin(a) : fn == 1 ? dmsa_det::det_cos(a) : fn == 2 ? dmsa_det::det_acos(a) : dmsa_det::det_atan2(y[i], a);
}
void launch_detmath_eval(int fn, const double* x, const double* y, int64_t n, double* out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_detmath_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fn, x, y, n, out);
}

void launch_window_pose_tables(const double* ctrl, const double* stamps, const double* fh_w, const double* traj_time, int B, int C, int n_t,
                               float* tables, float* tablesT, hipStream_t s, uint32_t* rot_same) {
    hipLaunchKernelGGL(k_window_pose_tables, dim3((n_t + 1 + 255) / 256, B), dim3(256), 0, s, ctrl, stamps, fh_w, traj_time, C, n_t, tables, tablesT, rot_same);
}
void launch_keyframe_pose_tables(const double* frames, int B, int F, float* tables, float* tablesT, hipStream_t s, uint32_t* rot_same) {
    hipLaunchKernelGGL(k_keyframe_pose_tables, dim3((F + 1 + 255) / 256, B), dim3(256), 0, s, frames, F, tables, tablesT, rot_same);
}

// ------------------------------------------------------------------------------------------------------------
// K2 — PCL-exact voxel lattice
// ------------------------------------------------------------------------------------------------------------
// (a) axis-aligned bounds of every block of 1024 points: lets the sequential bounding-box logic skip whole blocks.
__global__ __launch_bounds__(256) void k_block_aabb(const float4* __restrict__ global, int64_t n, float* __restrict__ aabb, uint32_t* __restrict__ zero,
                                                    int zero_words) {
    __shared__ float s_red[4][6];
    if (blockIdx.x == 0)  // the per-iteration counters of the voxelisation (a separate memset would be one more dispatch in the chain)
        for (int i = threadIdx.x; i < zero_words; i += 256) zero[i] = 0u;
    const int64_t base = (int64_t)blockIdx.x * kAabbBlock;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < kAabbBlock / 256; ++k) {
        const int64_t i = base + threadIdx.x + 256 * k;
        if (i < n) {
            const float4 p = global[i];
            if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
                mn[0] = fminf(mn[0], p.x), mn[1] = fminf(mn[1], p.y), mn[2] = fminf(mn[2], p.z);
                mx[0] = fmaxf(mx[0], p.x), mx[1] = fmaxf(mx[1], p.y), mx[2] = fmaxf(mx[2], p.z);
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_allminf(mn[a]);
        mx[a] = wave_allmaxf(mx[a]);
    }
    if (lane == 0)
        for (int a = 0; a < 3; ++a) s_red[wave][a] = mn[a], s_red[wave][3 + a] = mx[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, s_red[w][threadIdx.x]) : fmaxf(v, s_red[w][threadIdx.x]);
        aabb[(size_t)blockIdx.x * 8 + threadIdx.x] = v;
    }
}
// K0 + (a) in one pass: updateGlobalPoints writes the global points and leaves the bounds of every 1024-point block behind, so the
// voxelisation does not read the 24 MB it was just handed again (same arithmetic as k_transform(_normals) and k_block_aabb).
template <bool kNormals>
__global__ __launch_bounds__(256) void k_transform_aabb(const float4* __restrict__ local, const float4* __restrict__ nlocal, const float4* __restrict__ table,
                                                        float4* __restrict__ global, float4* __restrict__ nglobal, int64_t n, float* __restrict__ aabb,
                                                        uint32_t* __restrict__ zero, int zero_words) {
    __shared__ float s_red[4][6];
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_words; i += 256) zero[i] = 0u;
    const int64_t base = (int64_t)blockIdx.x * kAabbBlock;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < kAabbBlock / 256; ++k) {
        const int64_t i = base + threadIdx.x + 256 * k;
        if (i < n) {
            const float4 p = local[i];
            const int row = __float_as_int(p.w);
            const float4 r0 = table[3 * row], r1 = table[3 * row + 1], r2 = table[3 * row + 2];
            const float3 g = apply_row3(r0, r1, r2, p.x, p.y, p.z);
            global[i] = make_float4(g.x, g.y, g.z, 1.0f);
            if (kNormals) {
                const float4 v = nlocal[i];
                nglobal[i] = make_float4(sum3f(r0.x * v.x, r0.y * v.y, r0.z * v.z), sum3f(r1.x * v.x, r1.y * v.y, r1.z * v.z),
                                         sum3f(r2.x * v.x, r2.y * v.y, r2.z * v.z), 0.0f);
            }
            if (isfinite(g.x) && isfinite(g.y) && isfinite(g.z)) {
                mn[0] = fminf(mn[0], g.x), mn[1] = fminf(mn[1], g.y), mn[2] = fminf(mn[2], g.z);
                mx[0] = fmaxf(mx[0], g.x), mx[1] = fmaxf(mx[1], g.y), mx[2] = fmaxf(mx[2], g.z);
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int a = 0; a < 3; ++a) {
        mn[a] = wave_allminf(mn[a]);
        mx[a] = wave_allmaxf(mx[a]);
    }
    if (lane == 0)
        for (int a = 0; a < 3; ++a) s_red[wave][a] = mn[a], s_red[wave][3 + a] = mx[a];
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_red[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = threadIdx.x < 3 ? fminf(v, s_red[w][threadIdx.x]) : fmaxf(v, s_red[w][threadIdx.x]);
        aabb[(size_t)blockIdx.x * 8 + threadIdx.x] = v;
    }
}
void launch_transform_aabb(const float4* local, const float4* nlocal, const float4* table, float4* global, float4* nglobal, int64_t n, float* aabb, void* zero,
                           size_t zero_bytes, hipStream_t s) {
    if (n <= 0) {
        if (zero_bytes) (void)hipMemsetAsync(zero, 0, zero_bytes, s);
        return;
    }
    const int nb = (int)((n + kAabbBlock - 1) / kAabbBlock);
    if (nlocal)
        hipLaunchKernelGGL(k_transform_aabb<true>, dim3(nb), dim3(256), 0, s, local, nlocal, table, global, nglobal, n, aabb, static_cast<uint32_t*>(zero),
                           (int)(zero_bytes / 4));
    else
        hipLaunchKernelGGL(k_transform_aabb<false>, dim3(nb), dim3(256), 0, s, local, nlocal, table, global, nglobal, n, aabb, static_cast<uint32_t*>(zero),
                           (int)(zero_bytes / 4));
}
void launch_block_aabb(const float4* global, int64_t n, float* aabb, void* zero, size_t zero_bytes, hipStream_t s) {
    if (n <= 0) {
        if (zero_bytes) (void)hipMemsetAsync(zero, 0, zero_bytes, s);
        return;
    }
    const int nb = (int)((n + kAabbBlock - 1) / kAabbBlock);
    hipLaunchKernelGGL(k_block_aabb, dim3(nb), dim3(256), 0, s, global, n, aabb, static_cast<uint32_t*>(zero), (int)(zero_bytes / 4));
}

// (b) the incremental bounding box of OctreePointCloud::adoptBoundingBoxToPoint, replayed by one workgroup per
// resolution: blocks whose bounds fit the current box cannot trigger an event and are skipped 256 at a time; a
// block that does not fit is searched in parallel for its first violating point, lane 0 applies PCL's growth loop
// in double arithmetic, and the block is re-examined until it fits.
struct LatticeRun {
    double mn[3], mx[3];
    int depth, defined, nev, status;
};

__device__ void lattice_adopt(LatticeRun& st, const float4 p, const double res, int64_t idx, LatticeTable* tab, uint32_t (*shifts)[3]) {
    const double eps = (double)FLT_EPSILON;
    const float pc[3] = {p.x, p.y, p.z};
    while (true) {
        bool lo[3], hi[3];
        for (int a = 0; a < 3; ++a) lo[a] = (double)pc[a] < st.mn[a], hi[a] = (double)pc[a] >= st.mx[a];
        if (!(lo[0] || lo[1] || lo[2] || hi[0] || hi[1] || hi[2] || !st.defined)) break;
        if (st.defined) {
            if (st.nev >= kMaxLatticeEvents || st.depth >= 21) {
                st.status = DMSA_ERR_DEPTH;
                return;
            }
            double side = (double)(1u << st.depth) * res;
            for (int a = 0; a < 3; ++a) {
                uint32_t sh = 0;
                if (!hi[a]) {
                    st.mn[a] -= side;
                    sh = 1u << st.depth;
                }
                shifts[st.nev][a] = sh;
            }
            st.depth += 1;
            side = (double)(1u << st.depth) * res - eps;
            for (int a = 0; a < 3; ++a) st.mx[a] = st.mn[a] + side;
            tab->event_idx[st.nev] = idx;
            st.nev += 1;
            for (int a = 0; a < 3; ++a) tab->mn[st.nev][a] = st.mn[a];
            tab->depth[st.nev] = st.depth;
        } else {
            for (int a = 0; a < 3; ++a) {
                st.mn[a] = (double)pc[a] - res / 2;
                st.mx[a] = (double)pc[a] + res / 2;
            }
            // getKeyBitSize()
            unsigned mk[3];
            for (int a = 0; a < 3; ++a) mk[a] = (unsigned)ceil((st.mx[a] - st.mn[a] - eps) / res);
            const unsigned mv = max(max(max(mk[0], mk[1]), mk[2]), 2u);
            st.depth = (int)max(min(32u, (unsigned)ceil(log((double)mv) / log(2.0) - eps)), 0u);
            const double side = (double)(1u << st.depth) * res;
            for (int a = 0; a < 3; ++a) {
                const double over = (side - (st.mx[a] - st.mn[a])) / 2.0;
                if (over > eps) {
                    st.mn[a] -= over;
                    st.mx[a] += over;
                }
            }
            st.defined = 1;
            tab->first_idx = idx;
            for (int a = 0; a < 3; ++a) tab->mn[0][a] = st.mn[a];
            tab->depth[0] = st.depth;
        }
    }
}

constexpr int kLatticeThreads = 1024;            // = kAabbBlock: one point of the replayed block per thread
constexpr int kLatticeLdsBlocks = 2048;           // block bounds kept in LDS (48 KB); larger clouds read the rest from global memory
__global__ __launch_bounds__(kLatticeThreads) void k_lattice(const float4* __restrict__ global, int64_t n, const float* __restrict__ aabb, int nb, double res0,
                                                            double res1, int compress, LatticeTable* __restrict__ tables, uint32_t* __restrict__ sort_header0,
                                                            uint32_t* __restrict__ sort_header1, uint32_t* done /* dev_sync.h: both workgroups add one */,
                                                            int use_hint /* the table still holds the previous voxelisation's events: verify them first */) {
    static_assert(kLatticeThreads == kAabbBlock, "one thread per point of a block");
#ifdef DMSA_LATTICE_TIMING  // experiment build: phase stamps (100 MHz ticks since the kernel's start) in the unused tail of the table's mn array
    const long long lt0 = wall_clock64();
#define LAT_STAMP(k) if (threadIdx.x == 0) tables[blockIdx.x].mn[kMaxLatticeEvents - 8 + (k)][0] = (double)(wall_clock64() - lt0);
#else
#define LAT_STAMP(k)
#endif
    {  // the key kernels that follow count the sort digits into these headers
        uint32_t* h = blockIdx.x == 0 ? sort_header0 : sort_header1;
        if (h != nullptr)
            for (unsigned i = threadIdx.x; i < sizeof(SortHeader) / 4; i += kLatticeThreads) h[i] = 0u;
    }
    constexpr int kWaves = kLatticeThreads / 64;
    __shared__ float s_glob[kWaves][6];
    __shared__ LatticeRun st;
    __shared__ uint32_t s_shift[kMaxLatticeEvents][3];
    __shared__ int s_red[2][kWaves];
    __shared__ float s_bb[kLatticeLdsBlocks][6];
    int red_phase = 0;
    LatticeTable* tab = tables + blockIdx.x;
    const double res = blockIdx.x == 0 ? res0 : res1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        st.defined = 0, st.depth = 0, st.nev = 0, st.status = 0;
        for (int a = 0; a < 3; ++a) st.mn[a] = 0.0, st.mx[a] = 0.0;
        if (!use_hint) tab->first_idx = -1;  // (with a hint the table is read first; the replay resets it if the hint fails)
    }
    // the recorded table of the previous voxelisation (use_hint), staged by the last waves together with the block bounds of the others: one
    // lane reading it entry by entry from global memory would pay a round trip each
    __shared__ int s_ok, s_E, s_nown;
    __shared__ long long s_ev[kMaxLatticeEvents + 1];               // [0] first finite point, [1 + e] trigger of event e
    __shared__ long long s_own[kMaxLatticeEvents + 1];              // distinct blocks that hold the first point or a trigger
    __shared__ float4 s_q[kMaxLatticeEvents + 1];
    __shared__ double s_bmn[kMaxLatticeEvents + 1][3], s_bmx[kMaxLatticeEvents + 1][3];  // box of epoch e (after e events)
    __shared__ double s_tmn[kMaxLatticeEvents + 1][3];
    __shared__ int s_tdepth[kMaxLatticeEvents + 1], s_hdr[3];
    __shared__ int s_e0[kMaxLatticeEvents + 2];                      // events [s_e0[k], s_e0[k + 1]) have their trigger in special block k
    __shared__ float s_fmn[kMaxLatticeEvents + 1][3], s_fmx[kMaxLatticeEvents + 1][3];  // the boxes rounded INWARD to float (a cheap sufficient test)
    if (use_hint && tid >= kLatticeThreads - 128 && tid - (kLatticeThreads - 128) <= kMaxLatticeEvents) {
        const int t = tid - (kLatticeThreads - 128);
        s_tdepth[t] = tab->depth[t];
        for (int a = 0; a < 3; ++a) s_tmn[t][a] = tab->mn[t][a];
        if (t < kMaxLatticeEvents) s_ev[1 + t] = tab->event_idx[t];
        if (t == 0) s_hdr[0] = tab->num_events, s_hdr[1] = tab->status, s_hdr[2] = tab->defined, s_ev[0] = tab->first_idx;
    }
    {  // stage the block bounds; bounds of all finite points (for the key-range compression)
        float g6[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int b2 = tid; b2 < nb; b2 += kLatticeThreads) {
            const float4 lo = *reinterpret_cast<const float4*>(aabb + (size_t)b2 * 8), hi = *reinterpret_cast<const float4*>(aabb + (size_t)b2 * 8 + 4);
            const float bb[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
            if (b2 < kLatticeLdsBlocks)
                for (int a = 0; a < 6; ++a) s_bb[b2][a] = bb[a];
            if (bb[0] <= bb[3])
                for (int a = 0; a < 3; ++a) g6[a] = fminf(g6[a], bb[a]), g6[3 + a] = fmaxf(g6[3 + a], bb[3 + a]);
        }
        for (int a = 0; a < 3; ++a) g6[a] = wave_allminf(g6[a]), g6[3 + a] = wave_allmaxf(g6[3 + a]);
        if (lane == 0)
            for (int a = 0; a < 6; ++a) s_glob[wave][a] = g6[a];
    }
    __syncthreads();
    // minimum of an int over the workgroup (every thread gets it).  ONE barrier: the slots alternate, and the call after next, which
    // writes this call's slots again, lies behind the next call's barrier.
    auto block_min = [&](int v) {
        v = wave_allmin(v);
        if (lane == 0) s_red[red_phase][wave] = v;
        __syncthreads();
        int m = s_red[red_phase][0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) m = min(m, s_red[red_phase][w]);
        red_phase ^= 1;
        return m;
    };
    // ---- fast path: are the events of the PREVIOUS voxelisation still what the replay would find? ----
    // Between two iterations of optimizeSet the points move by millimetres: the same point defines the box, the same points push it
    // outward.  Whether they do is checked in PARALLEL, which the replay itself cannot be: (1) one lane re-applies PCL's growth steps to the
    // recorded trigger points (the first finite point, then event by event) and compares every box with the recorded one bit for bit;
    // (2) every other finite point must fit the box in force when it is inserted -- whole 1024-point blocks through their bounds, the few
    // blocks that hold a trigger point by point.  By induction over the insertion order the replay then produces exactly this table.
    // Any mismatch (another problem, moved events, garbage in a fresh buffer) falls through to the replay below.
    LAT_STAMP(0)
    constexpr int kHintBlocks = 12;  // blocks checked point by point (their points are fetched while the one lane works): more -> replay
    bool verified = false;
    if (use_hint) {
        if (tid == 0) {
            const int E = s_hdr[0];
            const long long first = s_ev[0];
            int ok = s_hdr[1] == 0 && s_hdr[2] == 1 && E >= 0 && E <= kMaxLatticeEvents && first >= 0 && first < n && n < (1ll << 31) - kAabbBlock;
            int nown = 0;
            if (ok) {
                long long prev = first;
                s_own[nown] = first / kAabbBlock, s_e0[nown] = 0, nown += 1;
                for (int e = 0; e < E; ++e) {
                    const long long ie = s_ev[1 + e];
                    ok = ok && ie > first && ie >= prev && ie < n;
                    if (ok && ie / kAabbBlock != s_own[nown - 1]) s_own[nown] = ie / kAabbBlock, s_e0[nown] = e, nown += 1;  // (ascending: equal blocks are neighbours)
                    prev = ie;
                }
                s_e0[nown] = E;
            }
            ok = ok && nown <= kHintBlocks;
            s_ok = ok, s_E = ok ? E : 0, s_nown = ok ? nown : 0;
        }
        __syncthreads();
        LAT_STAMP(1)
        // everybody's loads first: the trigger points for the one lane, one point of every special block for each thread
        if (s_ok && tid <= s_E) s_q[tid] = global[s_ev[tid]];
        float4 own_pt[kHintBlocks];
        const int nown = s_nown;
#pragma unroll
        for (int k = 0; k < kHintBlocks; ++k) {
            const long long pi = k < nown ? s_own[k] * kAabbBlock + tid : n;
            own_pt[k] = pi < n ? global[pi] : make_float4(NAN, NAN, NAN, 0.0f);  // (a non-finite point is skipped like in the replay)
        }
        __syncthreads();
        LAT_STAMP(2)
        if (s_ok && tid == 0) {
            const int E = s_E;
            const double eps = (double)FLT_EPSILON;
            LatticeRun c;
            c.defined = 0, c.depth = 0, c.nev = 0, c.status = 0;
            int ok = 1;
            {   // the first finite point defines the box (lattice_adopt's first branch, same operations)
                const float4 q = s_q[0];
                const float pc[3] = {q.x, q.y, q.z};
                ok = ok && isfinite(q.x) && isfinite(q.y) && isfinite(q.z);
                for (int a = 0; a < 3; ++a) c.mn[a] = (double)pc[a] - res / 2, c.mx[a] = (double)pc[a] + res / 2;
                unsigned mk[3];
                for (int a = 0; a < 3; ++a) mk[a] = (unsigned)ceil((c.mx[a] - c.mn[a] - eps) / res);
                const unsigned mv = max(max(max(mk[0], mk[1]), mk[2]), 2u);
                c.depth = (int)max(min(32u, (unsigned)ceil(log((double)mv) / log(2.0) - eps)), 0u);
                const double side = (double)(1u << c.depth) * res;
                for (int a = 0; a < 3; ++a) {
                    const double over = (side - (c.mx[a] - c.mn[a])) / 2.0;
                    if (over > eps) c.mn[a] -= over, c.mx[a] += over;
                }
                c.defined = 1;
                for (int a = 0; a < 3; ++a) ok = ok && __double_as_longlong(c.mn[a]) == __double_as_longlong(s_tmn[0][a]);
                ok = ok && c.depth == s_tdepth[0];
                for (int a = 0; a < 3; ++a) s_bmn[0][a] = c.mn[a], s_bmx[0][a] = c.mx[a];
            }
            for (int e = 0; e < E && ok; ++e) {  // one growth step per recorded event, on its recorded trigger point
                const float4 q = s_q[1 + e];
                const float pc[3] = {q.x, q.y, q.z};
                ok = ok && isfinite(q.x) && isfinite(q.y) && isfinite(q.z);
                bool lo[3], hi[3], viol = false;
                for (int a = 0; a < 3; ++a) lo[a] = (double)pc[a] < c.mn[a], hi[a] = (double)pc[a] >= c.mx[a], viol = viol || lo[a] || hi[a];
                ok = ok && viol && c.depth < 21;
                double side = (double)(1u << c.depth) * res;
                for (int a = 0; a < 3; ++a) {
                    uint32_t sh = 0;
                    if (!hi[a]) c.mn[a] -= side, sh = 1u << c.depth;
                    s_shift[e][a] = sh;
                }
                c.depth += 1;
                side = (double)(1u << c.depth) * res - eps;
                for (int a = 0; a < 3; ++a) c.mx[a] = c.mn[a] + side;
                for (int a = 0; a < 3; ++a) ok = ok && __double_as_longlong(c.mn[a]) == __double_as_longlong(s_tmn[e + 1][a]);
                ok = ok && c.depth == s_tdepth[e + 1];
                for (int a = 0; a < 3; ++a) s_bmn[e + 1][a] = c.mn[a], s_bmx[e + 1][a] = c.mx[a];
                if (e + 1 == E || s_ev[2 + e] != s_ev[1 + e]) {  // the trigger's last event: now it fits (else the replay would grow once more)
                    for (int a = 0; a < 3; ++a) ok = ok && !((double)pc[a] < c.mn[a]) && !((double)pc[a] >= c.mx[a]);
                }
            }
            s_ok = ok;
            if (ok) {
                st.defined = 1, st.depth = c.depth, st.nev = E, st.status = 0;
                for (int a = 0; a < 3; ++a) st.mn[a] = c.mn[a], st.mx[a] = c.mx[a];
            }
        }
        __syncthreads();
        if (s_ok && tid < 3 * (s_E + 1)) {  // the boxes rounded inward to float, one entry per thread
            const int e = tid / 3, a = tid - 3 * e;
            s_fmn[e][a] = __double2float_ru(s_bmn[e][a]), s_fmx[e][a] = __double2float_rd(s_bmx[e][a]);
        }
        __syncthreads();
        LAT_STAMP(3)
        if (s_ok) {
            const int E = s_E;
            const long long first = s_ev[0];
            int bad = 0;
            // A point or a block fits box e iff lo >= mn and hi < mx in DOUBLE (the replay's comparisons).  First in float against the box
            // rounded inward -- lo >= ru(mn) implies lo >= mn, hi < rd(mx) implies hi < mx -- and only what fails that goes through the
            // doubles: points within a float step of a face.
            auto fits = [&](const float lo3[3], const float hi3[3], int e) {
                bool f = true;
                for (int a = 0; a < 3; ++a) f = f && lo3[a] >= s_fmn[e][a] && hi3[a] < s_fmx[e][a];
                if (f) return true;
                f = true;
                for (int a = 0; a < 3; ++a) f = f && !((double)lo3[a] < s_bmn[e][a]) && !((double)hi3[a] >= s_bmx[e][a]);
                return f;
            };
            const int first32 = (int)first;
            for (int blk = tid; blk < nb; blk += kLatticeThreads) {  // whole blocks through their bounds
                float bb[6];
                if (blk < kLatticeLdsBlocks) {
                    for (int a = 0; a < 6; ++a) bb[a] = s_bb[blk][a];
                } else {
                    for (int a = 0; a < 6; ++a) bb[a] = aabb[(size_t)blk * 8 + a];
                }
                if (!(bb[0] <= bb[3])) continue;  // no finite point
                // epoch of a block that holds no trigger: events triggered in earlier blocks; a block that holds one is checked point by point
                int e = E;
                bool special = false;
                for (int k = nown - 1; k >= 0; --k) {
                    const int ob = (int)s_own[k];
                    special = special || ob == blk;
                    if (ob > blk) e = s_e0[k];
                }
                if (special) continue;
                if ((blk + 1) * kAabbBlock - 1 < first32)
                    bad = 1;  // a finite point in front of the recorded first one
                else if (!fits(bb, bb + 3, e))
                    bad = 1;
            }
            // the blocks that hold the first point or a trigger: point by point, from registers; epoch of a point = the events before its block
            // + the events of its block with a smaller index (usually one)
            // (Shortcut: a point inside EVERY box that is in force somewhere in its block -- the intersection of the inward-rounded boxes of the
            // epochs e0 .. e1 -- is inside the box of its own epoch, whichever that is; the bounds are uniform over the block, six float compares
            // settle all but the points around a trigger.  The intersection, not the first box: a box that grows downward keeps its maximum only
            // up to FLT_EPSILON (max = min + side - eps, PCL's adoptBoundingBoxToPoint), so "inside the first box" does not quite imply "inside
            // the later ones" -- round 5's shortcut tested the first box only.)
#pragma unroll
            for (int k = 0; k < kHintBlocks; ++k) {
                if (k < nown) {
                    const float4 q = own_pt[k];
                    const int base = (int)s_own[k] * kAabbBlock, e0 = s_e0[k], e1 = s_e0[k + 1];
                    float imn[3] = {s_fmn[e0][0], s_fmn[e0][1], s_fmn[e0][2]}, imx[3] = {s_fmx[e0][0], s_fmx[e0][1], s_fmx[e0][2]};
                    for (int e = e0 + 1; e <= e1; ++e)
                        for (int a = 0; a < 3; ++a) imn[a] = fmaxf(imn[a], s_fmn[e][a]), imx[a] = fminf(imx[a], s_fmx[e][a]);
                    const bool inside0 = base + tid > first32 && q.x >= imn[0] && q.x < imx[0] && q.y >= imn[1] && q.y < imx[1] && q.z >= imn[2] && q.z < imx[2];
                    if (!inside0 && isfinite(q.x) && isfinite(q.y) && isfinite(q.z)) {
                        int ep = e0;
                        bool trig = base + tid == first32;
                        for (int e = e0; e < e1; ++e) {
                            const int le = (int)s_ev[1 + e] - base;
                            ep += le < tid ? 1 : 0, trig = trig || le == tid;
                        }
                        const float c3[3] = {q.x, q.y, q.z};
                        if (base + tid < first32)
                            bad = 1;
                        else if (!trig && !fits(c3, c3, ep))  // (the triggers were checked by the one lane)
                            bad = 1;
                    }
                }
            }
            verified = block_min(bad ? 0 : 1) == 1;
        }
        LAT_STAMP(4)
        __syncthreads();
        if (!verified && tid == 0) {  // the replay starts from nothing
            st.defined = 0, st.depth = 0, st.nev = 0, st.status = 0;
            for (int a = 0; a < 3; ++a) st.mn[a] = 0.0, st.mx[a] = 0.0;
            tab->first_idx = -1;
        }
        __syncthreads();
    }
    LAT_STAMP(5)
    int cursor = verified ? nb : 0;
    while (cursor < nb) {
        // first block >= cursor whose bounds do not fit the current box
        int found = INT_MAX;
        for (int base = cursor; base < nb; base += kLatticeThreads) {
            const int blk = base + tid;
            int cand = INT_MAX;
            if (blk < nb) {
                float bb[6];
                if (blk < kLatticeLdsBlocks) {
                    for (int a = 0; a < 6; ++a) bb[a] = s_bb[blk][a];
                } else {
                    for (int a = 0; a < 6; ++a) bb[a] = aabb[(size_t)blk * 8 + a];
                }
                if (bb[0] <= bb[3]) {  // the block holds a finite point
                    bool viol = !st.defined;
                    for (int a = 0; a < 3; ++a) viol = viol || (double)bb[a] < st.mn[a] || (double)bb[3 + a] >= st.mx[a];
                    if (viol) cand = blk;
                }
            }
            const int m = block_min(cand);
            if (m != INT_MAX) {
                found = m;
                break;
            }
        }
        if (found == INT_MAX) break;
        // replay the block until none of its points violates the box; its points stay in registers, one per thread, and a point that
        // fitted once fits for good (the box only grows), so the search resumes behind the last pick
        const int64_t pi = (int64_t)found * kAabbBlock + tid;
        float4 p = float4{0.0f, 0.0f, 0.0f, 0.0f};
        bool live = false;
        if (pi < n) {
            p = global[pi];
            live = isfinite(p.x) && isfinite(p.y) && isfinite(p.z);
        }
        int from = 0;
        while (true) {
            int cand = INT_MAX;
            if (live && tid >= from) {
                bool viol = !st.defined;
                viol = viol || (double)p.x < st.mn[0] || (double)p.x >= st.mx[0];
                viol = viol || (double)p.y < st.mn[1] || (double)p.y >= st.mx[1];
                viol = viol || (double)p.z < st.mn[2] || (double)p.z >= st.mx[2];
                if (viol) cand = tid;
            }
            const int m = block_min(cand);
            if (m == INT_MAX) break;
            if (tid == m) lattice_adopt(st, p, res, pi, tab, s_shift);  // the violating thread applies PCL's growth loop itself
            __syncthreads();
            if (st.status != 0) break;
            from = m + 1;
        }
        if (st.status != 0) break;
        cursor = found + 1;
    }
    __syncthreads();
    LAT_STAMP(6)
    if (tid == 0) {
        tab->num_events = st.nev;
        tab->final_depth = st.depth;
        tab->status = st.status;
        tab->defined = st.defined;
        for (int a = 0; a < 3; ++a) tab->final_mn[a] = st.mn[a];
        // range of final keys per axis from the global bounds (one cell of slack on both sides: keys of earlier epochs were
        // rounded against a different origin), common prefix -> number of varying low bits
        int total = 0;
        for (int a = 0; a < 3; ++a) {
            float lo = s_glob[0][a], hi = s_glob[0][3 + a];
            for (int w = 1; w < kWaves; ++w) lo = fminf(lo, s_glob[w][a]), hi = fmaxf(hi, s_glob[w][3 + a]);
            int bits = st.depth;
            uint32_t base = 0;
            if (st.defined && lo <= hi) {
                const long long maxk = (1ll << st.depth) - 1;
                long long klo = (long long)floor(((double)lo - st.mn[a]) / res) - 1, khi = (long long)floor(((double)hi - st.mn[a]) / res) + 1;
                klo = klo < 0 ? 0 : klo, khi = khi > maxk ? maxk : khi;
                bits = 0;
                while (bits < st.depth && (klo >> bits) != (khi >> bits)) ++bits;
                base = (uint32_t)(klo >> bits);
            }
            tab->nbits[a] = bits, tab->key_base[a] = base;
            total += bits;
        }
        tab->compressed = compress, tab->out_of_range = 0, tab->code_or = 0;
        tab->pad3 = verified ? 1 : 0;  // (telemetry: the previous voxelisation's events were verified instead of replayed)
        tab->total_bits = total;
        uint32_t acc[3] = {0, 0, 0};
        for (int a = 0; a < 3; ++a) tab->suffix_shift[st.nev][a] = 0;
        for (int e = st.nev - 1; e >= 0; --e)
            for (int a = 0; a < 3; ++a) {
                acc[a] += s_shift[e][a];
                tab->suffix_shift[e][a] = acc[a];
            }
    }
    LAT_STAMP(7)
    if (done != nullptr) {  // the key kernels of the other level run on another stream
        __syncthreads();
        if (threadIdx.x == 0) dev_sync_signal(done);
    }
}
void launch_lattice(const float4* global, int64_t n, const float* aabb, int nb, double res0, double res1, bool compress, LatticeTable* tables,
                    void* sort_header0, void* sort_header1, hipStream_t s, uint32_t* done, bool use_hint) {
    hipLaunchKernelGGL(k_lattice, dim3(2), dim3(kLatticeThreads), 0, s, global, n, aabb, nb, res0, res1, compress ? 1 : 0, tables,
                       static_cast<uint32_t*>(sort_header0), static_cast<uint32_t*>(sort_header1), done, use_hint ? 1 : 0);
}

// (c) genOctreeKeyforPoint with the bounding box in force when the point was inserted, plus the integer shifts of
// later re-rootings; leaf code = x-major bit interleave at the final depth (depth-first leaf order).
__device__ __forceinline__ uint64_t spread3(uint32_t v) {  // 21 bits -> every third bit
    uint64_t x = v & 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

template <typename KeyT>  // uint32_t when the leaf code + invalid bit fit 32 bits (tree depth <= 10), else uint64_t
__global__ __launch_bounds__(256) void k_voxel_keys(const float4* __restrict__ global, int64_t n, LatticeTable* __restrict__ table, double res,
                                                    KeyT* __restrict__ code, uint32_t* __restrict__ idx, uint64_t code_or, SortHeader* __restrict__ sort_header,
                                                    uint32_t* __restrict__ sort_state, size_t sort_state_words, int sort_passes, uint32_t sort_last_mask) {
    __shared__ LatticeTable t;
    __shared__ uint32_t s_h[kSortMaxPasses][kSortBins];
    const bool counting = sort_header != nullptr && sizeof(KeyT) == 4;  // the sort that follows takes its digit histograms from here
    if (counting) {
        for (int p = 0; p < kSortMaxPasses; ++p) s_h[p][threadIdx.x] = 0u;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < sort_state_words; i += (size_t)gridDim.x * blockDim.x) sort_state[i] = 0u;
    }
    {
        const int words = sizeof(LatticeTable) / 4;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(table);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&t);
        for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) t.code_or = code_or;                         // the tag of this launch, whatever an earlier launch left
    if (blockIdx.x == 0 && threadIdx.x == 0) table->code_or = code_or;  // later kernels compare against lattice_invalid_code(*table)
    __syncthreads();
    const int nev = t.num_events;
    const uint64_t invalid = lattice_invalid_code(t);
    const int nx = t.nbits[0], ny = t.nbits[1], nz = t.nbits[2];
    const int maxb = max(nx, max(ny, nz));
    const int64_t last_ev = nev > 0 ? t.event_idx[nev - 1] : -1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 p = global[i];
        uint64_t c = invalid;
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            int e = nev;
            if (i < last_ev) {
                e = 0;
                while (e < nev && t.event_idx[e] <= i) ++e;
            }
            const uint32_t mask = (1u << t.depth[e]) - 1u;
            const uint32_t kx = ((uint32_t)(((double)p.x - t.mn[e][0]) / res) & mask) + t.suffix_shift[e][0];
            const uint32_t ky = ((uint32_t)(((double)p.y - t.mn[e][1]) / res) & mask) + t.suffix_shift[e][1];
            const uint32_t kz = ((uint32_t)(((double)p.z - t.mn[e][2]) / res) & mask) + t.suffix_shift[e][2];
            if (t.compressed) {
                // low nbits of every axis, interleaved level by level from the top (x, y, z order inside a level): same order as
                // the full depth-first code because the dropped high bits are identical for all points
                if ((kx >> nx) != t.key_base[0] || (ky >> ny) != t.key_base[1] || (kz >> nz) != t.key_base[2]) table->out_of_range = 1;
                uint64_t cc = 0;
                for (int l = maxb - 1; l >= 0; --l) {
                    if (l < nx) cc = (cc << 1) | ((kx >> l) & 1u);
                    if (l < ny) cc = (cc << 1) | ((ky >> l) & 1u);
                    if (l < nz) cc = (cc << 1) | ((kz >> l) & 1u);
                }
                c = cc | code_or;
            } else {
                c = (spread3(kx) << 2) | (spread3(ky) << 1) | spread3(kz) | code_or;
            }
        }
        code[i] = (KeyT)c;
        idx[i] = (uint32_t)i;
        if (counting) sort_hist_add(s_h, sort_passes, sort_last_mask, (uint32_t)c, true);
    }
    if (counting) {
        __syncthreads();
        for (int p = 0; p < sort_passes; ++p) {
            const uint32_t v = s_h[p][threadIdx.x];
            if (v) atomicAdd(&sort_header->hist[p][threadIdx.x], v);
        }
    }
}
void launch_voxel_keys(const float4* global, int64_t n, LatticeTable* table, double res, void* code, bool key32, uint32_t* idx, uint64_t code_or,
                       const SortPlan* sort, hipStream_t s) {
    if (n <= 0) return;
    SortHeader* h = sort && key32 ? sort->header : nullptr;
    if (key32)
        hipLaunchKernelGGL(k_voxel_keys<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, global, n, table, res, (uint32_t*)code, idx, code_or, h,
                           h ? sort->tile_state : nullptr, h ? sort->state_words : (size_t)0, h ? sort->passes : 0, h ? sort->last_mask : 255u);
    else
        hipLaunchKernelGGL(k_voxel_keys<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, global, n, table, res, (uint64_t*)code, idx, code_or,
                           (SortHeader*)nullptr, (uint32_t*)nullptr, (size_t)0, 0, 255u);
}

// ------------------------------------------------------------------------------------------------------------
// segmentation of the sorted arrays into leaves, acceptance, member gather
// ------------------------------------------------------------------------------------------------------------
// debug switch voxel_coherence: how many points changed their leaf code since the previous voxelisation (and keep this one's codes)
template <typename KeyT>
__global__ __launch_bounds__(256) void k_count_code_changes(const KeyT* __restrict__ now, KeyT* __restrict__ prev, int64_t n, int compare, unsigned long long* __restrict__ count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool diff = false;
    if (i < n) {
        const KeyT c = now[i];
        diff = compare && prev[i] != c;
        prev[i] = c;
    }
    const unsigned long long m = __ballot(diff);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}
void launch_count_code_changes(const void* now, void* prev, bool key32, int64_t n, bool compare, unsigned long long* count, hipStream_t s) {
    if (n <= 0) return;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (key32)
        hipLaunchKernelGGL(k_count_code_changes<uint32_t>, grid, dim3(256), 0, s, (const uint32_t*)now, (uint32_t*)prev, n, compare ? 1 : 0, count);
    else
        hipLaunchKernelGGL(k_count_code_changes<uint64_t>, grid, dim3(256), 0, s, (const uint64_t*)now, (uint64_t*)prev, n, compare ? 1 : 0, count);
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_head_flags(const KeyT* __restrict__ code, int64_t n, const LatticeTable* __restrict__ table,
                                                    int32_t* __restrict__ head) {
    const KeyT invalid = (KeyT)lattice_invalid_code(*table);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const KeyT c = code[i];
        head[i] = (c != invalid && (i == 0 || code[i - 1] != c)) ? 1 : 0;
    }
}
void launch_head_flags(const void* code_sorted, bool key32, int64_t n, const LatticeTable* table, int32_t* head, hipStream_t s) {
    if (n <= 0) return;
    if (key32)
        hipLaunchKernelGGL(k_head_flags<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint32_t*)code_sorted, n, table, head);
    else
        hipLaunchKernelGGL(k_head_flags<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, (const uint64_t*)code_sorted, n, table, head);
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_leaf_starts(const int32_t* __restrict__ head, const int32_t* __restrict__ leaf_incl,
                                                     const KeyT* __restrict__ code, const LatticeTable* __restrict__ table, int64_t n,
                                                     int32_t* __restrict__ leaf_start, LevelCounts* __restrict__ counts) {
    const KeyT invalid = (KeyT)lattice_invalid_code(*table);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (head[i]) leaf_start[leaf_incl[i] - 1] = (int32_t)i;
        if (code[i] != invalid && (i == n - 1 || code[i + 1] == invalid)) {
            counts->num_leaves = leaf_incl[i];
            leaf_start[leaf_incl[i]] = (int32_t)(i + 1);
        }
    }
}
void launch_leaf_starts(const int32_t* head, const int32_t* leaf_of_pos, const void* code_sorted, bool key32, const LatticeTable* table, int64_t n,
                        int32_t* leaf_start, LevelCounts* counts, hipStream_t s) {
    if (n <= 0) return;
    if (key32)
        hipLaunchKernelGGL(k_leaf_starts<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, head, leaf_of_pos, (const uint32_t*)code_sorted, table, n,
                           leaf_start, counts);
    else
        hipLaunchKernelGGL(k_leaf_starts<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, head, leaf_of_pos, (const uint64_t*)code_sorted, table, n,
                           leaf_start, counts);
}

// head flags + inclusive scan + leaf starts in ONE pass over the sorted codes (the three-kernel form above remains for the static
// map).  Single-pass chained scan: tiles of 8192 positions are handed out by an atomic ticket, a tile publishes its number of leaf
// heads and looks back over the earlier tiles.  Nothing is cleared between calls: every published word carries the call's epoch
// (other epochs read as "not there yet") and the ticket counter runs on, the host passing the value it had when the call started.
constexpr int kSegThreads = 512;
// positions per thread of a tile: small inputs get small tiles (a window of 25 000 points would be four tiles of 8192 on four compute units)
static inline int seg_items_for(int64_t n) { return n <= (1 << 16) ? 2 : n <= (1 << 18) ? 4 : 16; }
template <typename KeyT, int kSegItems>
__global__ __launch_bounds__(kSegThreads) void k_leaf_segments(const KeyT* __restrict__ code, int64_t n, const LatticeTable* __restrict__ table,
                                                                int32_t* __restrict__ leaf_incl, int32_t* __restrict__ leaf_start,
                                                                LevelCounts* __restrict__ counts, unsigned long long* __restrict__ state /* [0]: ticket, [1 + tile] */,
                                                                uint32_t epoch, uint32_t ticket_base) {
    constexpr int kSegTile = kSegThreads * kSegItems;
    __shared__ uint32_t s_tile;
    __shared__ int s_wave_total[kSegThreads / 64];
    __shared__ int s_excl;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_tile = atomicAdd(reinterpret_cast<unsigned int*>(state), 1u) - ticket_base;
    __syncthreads();
    const uint32_t tile = s_tile;
    const KeyT invalid = (KeyT)lattice_invalid_code(*table);
    const int64_t base = (int64_t)tile * kSegTile;
    int flag[kSegItems], incl[kSegItems];
    bool last_valid[kSegItems];
    int run = 0;  // heads in the earlier rows of this wave
#pragma unroll
    for (int k = 0; k < kSegItems; ++k) {
        const int64_t i = base + (int64_t)(wave * kSegItems + k) * 64 + lane;
        flag[k] = 0, last_valid[k] = false;
        if (i < n) {
            const KeyT c = code[i];
            flag[k] = (c != invalid && (i == 0 || code[i - 1] != c)) ? 1 : 0;
            last_valid[k] = c != invalid && (i == n - 1 || code[i + 1] == invalid);
        }
        const int sc = wave_incl_scan_dpp(flag[k]);
        incl[k] = run + sc;
        run += __builtin_amdgcn_readlane(sc, 63);
    }
    if (lane == 0) s_wave_total[wave] = run;
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSegThreads / 64; ++w) {
        const int t = s_wave_total[w];
        if (w < wave) before += t;
        total += t;
    }
    if (wave == 0) {
        // look-back by one wave: lane u inspects predecessor t - u, 64 tiles per round trip (a round trip to the device-coherent level
        // of the cache hierarchy costs about a microsecond on this multi-die GPU, so the chain is walked as wide as possible)
        constexpr unsigned long long kPartial = 1ull << 30, kPrefix = 2ull << 30;
        const unsigned long long tag = (unsigned long long)epoch << 32;
        unsigned long long* st = state + 1;
        int excl = 0;
        if (tile == 0) {
            if (lane == 0) __hip_atomic_store(st, tag | kPrefix | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (lane == 0) __hip_atomic_store(st + tile, tag | kPartial | (unsigned)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t t = (int64_t)tile - 1;
            while (true) {
                const int64_t mine = t - lane;
                const unsigned long long v = mine >= 0 ? __hip_atomic_load(st + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | kPrefix);
                const uint32_t f = (uint32_t)(v >> 32) == epoch ? ((uint32_t)v) >> 30 : 0u;
                const unsigned long long ready = __ballot(f != 0u), prefix = __ballot(f == 2u);
                // lanes 0 .. stop-1 are consumed: up to and including the first prefix, or up to the first entry that is not there yet
                const int first_missing = ready == ~0ull ? 64 : __builtin_ctzll(~ready);
                const int first_prefix = prefix == 0ull ? 64 : __builtin_ctzll(prefix);
                const bool done = first_prefix < first_missing;
                const int stop = done ? first_prefix + 1 : first_missing;
                int part = lane < stop ? (int)((uint32_t)v & ((1u << 30) - 1u)) : 0;
                part = wave_incl_scan_dpp(part);
                excl += __builtin_amdgcn_readlane(part, 63);
                if (done) break;
                t -= stop;
            }
            if (lane == 0) __hip_atomic_store(st + tile, tag | kPrefix | (unsigned)(excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0) s_excl = excl;
    }
    __syncthreads();
    const int off = s_excl + before;
#pragma unroll
    for (int k = 0; k < kSegItems; ++k) {
        const int64_t i = base + (int64_t)(wave * kSegItems + k) * 64 + lane;
        if (i < n) {
            const int li = off + incl[k];
            leaf_incl[i] = li;
            if (flag[k]) leaf_start[li - 1] = (int32_t)i;
            if (last_valid[k]) {
                counts->num_leaves = li;
                leaf_start[li] = (int32_t)(i + 1);
            }
        }
    }
}
void launch_leaf_segments(const void* code_sorted, bool key32, int64_t n, const LatticeTable* table, int32_t* leaf_incl, int32_t* leaf_start, LevelCounts* counts,
                          unsigned long long* state, uint32_t epoch, uint32_t ticket_base, hipStream_t s) {
    if (n <= 0) return;
    const unsigned tiles = (unsigned)leaf_segment_tiles(n);
#define DMSA_LAUNCH_SEG(KEY, ITEMS)                                                                                                                       \
    hipLaunchKernelGGL((k_leaf_segments<KEY, ITEMS>), dim3(tiles), dim3(kSegThreads), 0, s, (const KEY*)code_sorted, n, table, leaf_incl, leaf_start, counts, state, \
                       epoch, ticket_base)
    const int items = seg_items_for(n);
    if (key32) {
        if (items == 2)
            DMSA_LAUNCH_SEG(uint32_t, 2);
        else if (items == 4)
            DMSA_LAUNCH_SEG(uint32_t, 4);
        else
            DMSA_LAUNCH_SEG(uint32_t, 16);
    } else {
        if (items == 2)
            DMSA_LAUNCH_SEG(uint64_t, 2);
        else if (items == 4)
            DMSA_LAUNCH_SEG(uint64_t, 4);
        else
            DMSA_LAUNCH_SEG(uint64_t, 16);
    }
#undef DMSA_LAUNCH_SEG
}
int leaf_segment_tiles(int64_t n) {
    const int64_t tile = (int64_t)kSegThreads * seg_items_for(n);
    return (int)((n + tile - 1) / tile);
}

// DmsaOptimizer.h:302-307: a leaf becomes a point set iff size >= minNumberPts and max(id) != min(id)
__global__ __launch_bounds__(256) void k_leaf_accept(const int32_t* __restrict__ leaf_start, const uint32_t* __restrict__ idx_sorted,
                                                     const int32_t* __restrict__ ring, const LevelCounts* __restrict__ counts, int min_pts,
                                                     int64_t capacity, int32_t* __restrict__ slot_acc, int32_t* __restrict__ slot_cnt) {
    const int nl = counts->num_leaves;  // slots of leaves >= nl are never read (k_leaf_scan stops at nl)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; l < nl; l += stride) {
        int acc = 0, cnt = 0;
        {
            const int b = leaf_start[l], e = leaf_start[l + 1];
            cnt = e - b;
            if (cnt >= min_pts) {
                // Members are sorted by point index, and neighbouring points of a scan line share their ring id: the walk to the first
                // different id is tens of steps in the coarse leaves, each one two dependent loads.  Eight members per round, their
                // loads independent of each other (the answer is an OR, the order of the tests does not matter).
                const int first = ring[idx_sorted[b]];
                for (int j = b + 1; j < e && !acc; j += 8) {
                    uint32_t pi[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) pi[u] = idx_sorted[min(j + u, e - 1)];
                    int id[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) id[u] = ring[pi[u]];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc |= id[u] != first ? 1 : 0;
                }
            }
        }
        slot_acc[2 * l] = acc;
        slot_cnt[2 * l] = acc ? cnt : 0;
        slot_acc[2 * l + 1] = 0;
        slot_cnt[2 * l + 1] = 0;
    }
}
void launch_leaf_accept(const int32_t* leaf_start, const uint32_t* idx_sorted, const int32_t* ring, const LevelCounts* counts, int min_pts,
                        int64_t capacity, int32_t* slot_acc, int32_t* slot_cnt, hipStream_t s) {
    if (capacity <= 0) return;
    hipLaunchKernelGGL(k_leaf_accept, dim3(512), dim3(256), 0, s, leaf_start, idx_sorted, ring, counts, min_pts, capacity,
                       slot_acc, slot_cnt);
}

// The quadratic part of splitSet (Gaussians.h:36-51), position-parallel: every sorted position a of an accepted leaf scans
// all members c of its leaf (normals pre-gathered in sorted order, so a wave inside one big leaf reads the same address)
// and keeps its first best partner.  Work per leaf is n^2 / 64 wave-iterations spread over n / 64 waves instead of one.
// Work decomposition: a TASK is (wave block of 64 consecutive sorted positions) x (one chunk of the partner range).  The
// union of the leaves touched by the 64 positions is contiguous; it is cut into at most kSplitMaxChunks chunks of >=
// kSplitChunk partners, so a 17 000-point leaf becomes ~275 x 64 tasks that fill the chip, while the task list stays <= n
// entries.  k_split_tasks writes the list (slots reserved with one atomicAdd per wave block; the order is irrelevant),
// k_split_pairs walks it with persistent waves.
constexpr int kSplitChunk = 256;
constexpr int kSplitMaxChunks = 64;
constexpr int kSplitStripes = 256;   // ticket counters of k_split_pairs (= threads of a k_split_prepare workgroup, which clears them)
struct SplitTask {
    int32_t wave_block;  // positions 64*wave_block .. +63
    int32_t c0, c1;      // partner positions [c0, c1)
    int32_t pad;
};
__device__ __forceinline__ void split_position_range(int64_t i, int64_t nvalid, int nl, const int32_t* __restrict__ leaf_incl,
                                                     const int32_t* __restrict__ leaf_start, const int32_t* __restrict__ slot_acc, bool& active,
                                                     int& b, int& e) {
    active = i < nvalid;
    b = INT_MAX, e = 0;
    if (active) {
        const int l = leaf_incl[i] - 1;
        active = l >= 0 && l < nl && slot_acc[2 * l] != 0;
        if (active) b = leaf_start[l], e = leaf_start[l + 1];
    }
}
__global__ __launch_bounds__(1024) void k_split_tasks(const int32_t* __restrict__ leaf_incl, const int32_t* __restrict__ leaf_start,
                                                      const int32_t* __restrict__ slot_acc, int64_t n_valid_cap,
                                                      const LevelCounts* __restrict__ counts, SplitTask* __restrict__ tasks,
                                                      int32_t* __restrict__ num_tasks) {
    __shared__ int s_cnt[16], s_base;
    const int nl = counts->num_leaves;
    const int64_t nvalid = min((int64_t)leaf_start[nl], n_valid_cap);
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if ((int64_t)blockIdx.x * 1024 >= nvalid) return;
    bool active;
    int b, e;
    split_position_range(i, nvalid, nl, leaf_incl, leaf_start, slot_acc, active, b, e);
    int lo = b, hi = e;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) lo = min(lo, __shfl_xor(lo, m)), hi = max(hi, __shfl_xor(hi, m));
    if (hi > lo) lo &= ~63;  // partner blocks are the 64-aligned blocks of the sorted array (k_split_cones bounds them); positions in front of the leaf are masked
    // Only partners BEHIND a position are searched (k_split_pairs): |n_a + n_c| has the same bits as |n_c + n_a|, so the leaf's first minimal
    // pair in the reference's (a outer, c inner) order has a < c -- a minimal pair with c < a would make (c, a) an earlier one.  The partner
    // range of a wave block therefore starts at the block itself: half the pairs.
    lo = max(lo, (int)((i >> 6) << 6));
    const int len = max(0, hi - lo);
    int chunk = max(kSplitChunk, (len + kSplitMaxChunks - 1) / kSplitMaxChunks);
    chunk = (chunk + 63) & ~63;
    const int nchunks = (len + chunk - 1) / chunk;  // 0 for a wave block without accepted leaves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = nchunks;
    __syncthreads();
    if (threadIdx.x == 0) {  // one atomic per 1024 positions: the counter is a single contended address
        int tot = 0;
        for (int w = 0; w < 16; ++w) {
            const int c = s_cnt[w];
            s_cnt[w] = tot, tot += c;
        }
        s_base = tot ? atomicAdd(num_tasks, tot) : 0;
    }
    __syncthreads();
    if (lane < nchunks) {  // nchunks <= kSplitMaxChunks = 64
        SplitTask t;
        t.wave_block = (int32_t)(i >> 6), t.c0 = lo + lane * chunk, t.c1 = min(hi, lo + (lane + 1) * chunk), t.pad = 0;
        tasks[s_base + s_cnt[wave] + lane] = t;
    }
}
// Bounds that let k_split_pairs skip whole 64 x 64 blocks of pairs.  splitSet only ever uses the pair with the smallest |n_a + n_c|, and
// only if that is <= 0.5 (Gaussians.h:54): a pair whose norm is certainly above 0.5 can be left out without changing anything.  Members
// are sorted by point index, so 64 consecutive partners mostly lie on one surface: their normals fit a narrow cone (axis = normalised sum,
// half-angle phi).  For a position a at angle alpha from the axis every partner b of the block is at most min(pi, alpha + phi) away from
// a, hence |a + b|^2 >= |a|^2 + bmin^2 + 2 |a| bmax cos(min(pi, alpha + phi)) (the last term only counts when negative).  cone[blk] =
// (axis, cos phi), norms[blk] = (bmin, bmax); a block with a non-finite or vanishing sum gets cos phi = -1: never skipped.
__device__ __forceinline__ void split_block_cone(const float4 v, bool in, int lane, int64_t blk, float4* __restrict__ cone, float2* __restrict__ norms) {
    const float sx = wave_allsum(v.x), sy = wave_allsum(v.y), sz = wave_allsum(v.z);
    const float s2 = sum3f(sx * sx, sy * sy, sz * sz), nv = sqrtf(sum3f(v.x * v.x, v.y * v.y, v.z * v.z));
    float cx = 0.f, cy = 0.f, cz = 0.f, cosphi = -1.0f;
    float nmin = in ? nv : FLT_MAX, nmax = in ? nv : 0.0f;
    nmin = wave_allminf(nmin), nmax = wave_allmaxf(nmax);
    if (s2 > 1e-6f && s2 < 1e12f && nmin > 1e-3f) {  // (a NaN fails every comparison)
        const float inv = 1.0f / sqrtf(s2);
        cx = sx * inv, cy = sy * inv, cz = sz * inv;
        float d = in ? sum3f(v.x * cx, v.y * cy, v.z * cz) / nv : 1.0f;
        if (!(d >= -1.0f)) d = -1.0f;
        cosphi = fminf(wave_allminf(d), 1.0f) - 1e-5f;  // (rounding of the dot products: the cone a hair wider)
    }
    if (!(nmin <= nmax)) nmin = 0.0f, nmax = FLT_MAX, cosphi = -1.0f;
    if (lane == 0) cone[blk] = make_float4(cx, cy, cz, cosphi), norms[blk] = make_float2(nmin, nmax);
}
// Everything the pair search needs, in ONE pass over the sorted positions (four launches until round 4: gather, two fills, cones): the
// normals in sorted order (ring id in w), pair_best = "no partner", the task counter = 0, and the cone of every 64 consecutive normals.
__global__ __launch_bounds__(256) void k_split_prepare(const uint32_t* __restrict__ idx_sorted, const float4* __restrict__ nglobal,
                                                       const int32_t* __restrict__ ring, int64_t n, float4* __restrict__ nsorted,
                                                       unsigned long long* __restrict__ pair_best, int32_t* __restrict__ num_tasks,
                                                       float4* __restrict__ cone, float2* __restrict__ norms, int32_t* __restrict__ tickets) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;   // a multiple of 64: a wave always holds one aligned block of 64 positions
    if (blockIdx.x == 0 && threadIdx.x == 0) *num_tasks = 0;
    if (blockIdx.x == 0) tickets[threadIdx.x * 16] = 0;       // kSplitStripes = blockDim.x counters, 64 bytes apart
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n; base += stride) {
        const int64_t i = base + lane;
        const bool in = i < n;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in) {
            const uint32_t pi = idx_sorted[i];
            v = nglobal[pi];
            v.w = __int_as_float(ring[pi]);  // the normal's w is unused: carry the ring id in sorted order
            nsorted[i] = v;
            pair_best[i] = ~0ull;
        }
        split_block_cone(v, in, lane, base >> 6, cone, norms);
    }
}

// (Partners c <= a are never looked at: see k_split_tasks.  pair_best[a] is the first best partner BEHIND a.)
// A task's 64 positions meet its partners 64 at a time: one coalesced vector load puts partner c0 + lane into every lane
// (the next 64 are prefetched meanwhile), then the partners are broadcast lane by lane with v_readlane.  No LDS and no scalar
// cache in the loop (an LDS-staged loop was bound by LDS read bandwidth, a scalar-load loop by scalar-cache misses).  The
// chunks of a position are merged with a 64-bit atomicMin on (float bits of the distance, partner rank): distances are >= 0,
// so the unsigned order of the bits is the float order, and the smaller rank wins a tie -- the reference's first strict minimum.
__device__ __forceinline__ float bcast_lane(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
// lane K of every row of 16 lanes, for all lanes of that row (row_newbcast); as the source of an add the compiler folds it into the
// instruction (v_add_f32_dpp)
template <int K>
__device__ __forceinline__ float dpp_row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, false));
}
__global__ __launch_bounds__(256) void k_split_pairs(const int32_t* __restrict__ leaf_incl, const int32_t* __restrict__ leaf_start,
                                                     const int32_t* __restrict__ slot_acc, const float4* __restrict__ nsorted, int64_t n_valid_cap,
                                                     const LevelCounts* __restrict__ counts, const SplitTask* __restrict__ tasks,
                                                     const int32_t* __restrict__ num_tasks, unsigned long long* __restrict__ pair_best,
                                                     const float4* __restrict__ cone, const float2* __restrict__ norms, unsigned long long* __restrict__ skipped,
                                                     int32_t* __restrict__ tickets) {
    const int nl = counts->num_leaves;
    const int64_t nvalid = min((int64_t)leaf_start[nl], n_valid_cap);
    const int ntasks = *num_tasks;
    unsigned long long n_blocks = 0, n_skipped = 0;
    const int lane = threadIdx.x & 63;
    const float kFar = 3.0e19f;  // (n_a + kFar)^2 overflows to +inf: padding partners can never win
    // Tasks differ by a factor of ten in cost (a chunk of partners is either skipped block by block or searched pair by pair), so a fixed
    // share per wave left the chip waiting for the unlucky waves.  Stripe q holds tasks q, q + kSplitStripes, ... (neighbouring tasks are
    // alike, so the stripes get the same mix) and is handed out task by task, through its own counter (64 bytes apart: one counter for
    // all waves would serialise ~10^5 atomics), to the waves of the workgroups q, q + kSplitStripes, ...
    const int stripe = (int)(blockIdx.x % kSplitStripes);
    for (;;) {
        int t = 0;
        if (lane == 0) t = stripe + kSplitStripes * atomicAdd(&tickets[stripe * 16], 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntasks) break;
        const SplitTask task = tasks[t];
        const int64_t i = (int64_t)task.wave_block * 64 + lane;
        const int c0 = __builtin_amdgcn_readfirstlane(task.c0), c1 = __builtin_amdgcn_readfirstlane(task.c1);
        bool active;
        int b, e;
        split_position_range(i, nvalid, nl, leaf_incl, leaf_start, slot_acc, active, b, e);
        // What a pair has to beat (squared): splitSet only uses the leaf's smallest |n_a + n_c|, and only if it is <= 0.5 (Gaussians.h:54), so a
        // pair that is certainly above 0.5 need not be looked at.  0.2501 = 0.5^2 plus far more than the rounding of the bound and of the
        // reference's own norm.  (A bound that follows the smallest norm found in the leaf so far -- one word per leaf, atomicMin -- was
        // measured as well: 3.8 ms per voxelisation with one atomic per lane, 0.51 ms with one per wave, against 0.43 ms with this constant.)
        const float bound = 0.2501f;
        // every position may pair with the whole chunk: the chunk lies inside the position's leaf and BEHIND the position (see k_split_tasks)
        const bool uniform = __ballot(active && b <= c0 && e >= c1 && i < c0) == ~0ull;
        const float4 na = active ? nsorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        // The reference compares norms (sqrt) with a strict '<'.  sqrt is monotone, so a candidate whose SQUARED norm is not
        // below the best squared norm can never win; the (correctly rounded, expensive) sqrt is only taken for the rest.
        // best_sq starts AT the bound: a candidate above it changes nothing downstream (k_leaf_split only looks at minima <= 0.5), so the
        // ordered update below never runs for the ~parallel pairs that make up almost every block that is not skipped outright
        float best_sq = active ? bound : -1.0f;
        int best_c = INT_MAX;   // sorted position of the best partner so far
        const int self = (int)i;
        // The reference's update, in partner order: `norm < min` on the ROUNDED square roots.  Two squares more than 2^-21 apart (relative)
        // have square roots more than one float step apart, so their roundings compare like the squares: no sqrt on that branch.  Inside
        // the band the roots are taken and compared as the reference does (equal roots: the earlier partner stays).
        auto exact = [&](float sq, int c) {
            if (sq < best_sq && c != self) {
                if (sq < best_sq * 0.9999995f /* 1 - 2^-21 */ || sqrtf(sq) < sqrtf(best_sq)) best_sq = sq, best_c = c;
            }
        };
        struct Block {
            float4 row[4];   // partners 16 m + (lane & 15) of a block of 64, m = 0 .. 3: every row of 16 lanes holds the same sixteen
        };
        auto load_block = [&](int j) {
            Block blk;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int jj = j + 16 * m + (lane & 15);
                blk.row[m] = jj < c1 ? nsorted[jj] : make_float4(kFar, kFar, kFar, 0.f);
            }
            return blk;
        };
        // for the block bound (k_split_cones): |a|^2, 1 / |a|
        const float na2 = sum3f(na.x * na.x, na.y * na.y, na.z * na.z), na_len = sqrtf(na2), na_inv = 1.0f / na_len;
        Block cur = load_block(c0);
        for (int j = c0; j < c1; j += 64) {
            const Block nxt = load_block(j + 64);   // (beyond c1: padding, no load)
            {   // can any pair of (these 64 positions) x (this partner block) have a norm <= 0.5?  If certainly not: skip the 4096 pairs.
                const float4 cn = cone[j >> 6];
                const float2 nb = norms[j >> 6];
                const float ca = sum3f(na.x * cn.x, na.y * cn.y, na.z * cn.z) * na_inv;      // cos alpha
                // sin alpha must not be UNDERestimated: ca carries ~1e-7 of error (normalised axis x 1 / |a|), which near ca = 1 is up to 4.5e-4 in
                // sqrt(1 - ca^2) -- more than the slack of the bound.  1 - ca^2 is off by at most 2 |ca| 1e-7 + eps < 4e-7: add that before the root.
                const float sa = sqrtf(fmaxf(0.0f, 1.0f - ca * ca) + 4e-7f), sp = sqrtf(fmaxf(0.0f, 1.0f - cn.w * cn.w) + 4e-7f);
                const float cmin = ca <= -cn.w ? -1.0f : ca * cn.w - sa * sp - 2e-6f;        // cos(min(pi, alpha + phi)), rounded down
                const float lb = na2 + nb.x * nb.x + (cmin < 0.0f ? 2.0f * na_len * nb.y * cmin : 0.0f);
                // inactive lanes agree to anything
                const bool far = !active || (cn.w > -1.0f && lb * 0.99999f > bound);
                n_blocks += 1;
                if (__ballot(far) == ~0ull) {
                    n_skipped += 1;
                    cur = nxt;
                    continue;
                }
            }
            // (A pre-test on the inner product a.c -- three instructions per candidate instead of eight -- was measured: 135 -> 230 us.  The
            // blocks that survive the cone test are walls seen from both sides, where EVERY pair is within 1e-6 of the running minimum:
            // a conservative margin lets them all through and the exact sum is computed on top of the pre-test.)
            // The kernel is bound by vector instruction issue (93 % of the slots, scripts/pmc_kernel.sh), and a quarter of its instructions were
            // v_readlane broadcasts of the partners.  Now a partner reaches the lanes as a MODIFIER of the add that consumes it (row_newbcast:
            // lane k of a row of 16 to all lanes of the row); four loads put partners 16 m .. 16 m + 15 of the block into every row.  All
            // lanes still see the partners in ascending order, which the strict '<' of `exact` relies on.
            auto four = [&](auto kc, const float4& pm, int m) {  // partners 16 m + k .. + 3 of the block, k = kc.value
                constexpr int k = decltype(kc)::value;
                float q[4];
                auto one = [&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    const float sx = na.x + dpp_row_bcast<k + u>(pm.x), sy = na.y + dpp_row_bcast<k + u>(pm.y), sz = na.z + dpp_row_bcast<k + u>(pm.z);
                    q[u] = sum3f(sx * sx, sy * sy, sz * sz);
                    if (!uniform) q[u] = (j + 16 * m + k + u > max(b - 1, self) && j + 16 * m + k + u < e) ? q[u] : INFINITY;  // partner of another leaf, or not behind a
                };
                one(std::integral_constant<int, 0>{}), one(std::integral_constant<int, 1>{}), one(std::integral_constant<int, 2>{}), one(std::integral_constant<int, 3>{});
                if (fminf(fminf(q[0], q[1]), fminf(q[2], q[3])) < best_sq) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) exact(q[u], j + 16 * m + k + u);
                }
            };
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                four(std::integral_constant<int, 0>{}, cur.row[m], m), four(std::integral_constant<int, 4>{}, cur.row[m], m);
                four(std::integral_constant<int, 8>{}, cur.row[m], m), four(std::integral_constant<int, 12>{}, cur.row[m], m);
            }
            cur = nxt;
        }
        if (active && best_c != INT_MAX)
            atomicMin(&pair_best[i], ((unsigned long long)__float_as_uint(sqrtf(best_sq)) << 32) | (unsigned long long)(uint32_t)(best_c - b));
    }
    // statistics (debug counters): 64 stripes of two words, one pair of atomics per workgroup -- every wave adding to ONE pair of words cost
    // 0.4 ms per voxelisation
    if (skipped != nullptr) {
        __shared__ unsigned long long s_n[2];
        if (threadIdx.x == 0) s_n[0] = 0, s_n[1] = 0;
        __syncthreads();
        if (lane == 0 && n_blocks) atomicAdd(&s_n[0], n_blocks), atomicAdd(&s_n[1], n_skipped);
        __syncthreads();
        if (threadIdx.x == 0 && s_n[0]) {
            unsigned long long* dst = skipped + 2 * (blockIdx.x & 63);
            atomicAdd(&dst[0], s_n[0]), atomicAdd(&dst[1], s_n[1]);
        }
    }
}

// Gaussians.h:27-85 splitSet + the split branch of createGaussianSets (DmsaOptimizer.h:310-337).  One GROUP of kThreads
// threads per accepted leaf: a wave (kThreads = 64) for leaves of <= kSplitBigLeaf members, a 1024-thread workgroup for
// the few larger ones (a 17 000-point leaf would otherwise be walked by one wave).  pos_slot_rank[i] = rank within its
// set, sign bit set for the second set.
constexpr int kSplitBigLeaf = 1024;
template <int kThreads>
__device__ __forceinline__ void leaf_split_groups(int group, int ngroups, int tid, const int32_t* __restrict__ leaf_start, const float4* __restrict__ nsorted,
                                                  const LevelCounts* __restrict__ counts, int min_pts, const unsigned long long* __restrict__ pair_best,
                                                  int32_t* __restrict__ slot_acc, int32_t* __restrict__ slot_cnt, int32_t* __restrict__ pos_slot_rank) {
    constexpr int kWaves = kThreads / 64;
    __shared__ float s_best[kWaves];
    __shared__ long long s_pair[kWaves];
    __shared__ int s_c1[kWaves], s_c2[kWaves], s_mn[kWaves], s_mx[kWaves];   // (only the workgroup-wide groups, kWaves > 1, touch these)
    const int nl = counts->num_leaves;
    const int lane = tid & 63, wave = tid >> 6;
    // Which leaves are this group's?  Accepted ones of its size class -- a small minority of the fine level's ~10^5 leaves.  Asking leaf by
    // leaf (`if (!slot_acc[2 * l]) continue;`) is one dependent memory round trip per leaf and group: the 256 workgroup-wide groups each
    // walked nl / 256 leaves to find no big leaf at all, 100 us at the fine level while their 8192 waves held every wave slot of the chip
    // against the coarse level's pair search on the other stream.  Now a group looks at kThreads of its leaves per round trip -- one leaf per
    // thread, (accepted, begin, end) loaded together -- and only walks the ones that qualify.
    __shared__ unsigned long long s_mask[kWaves];
    auto process = [&](const int l, const int b, const int e) {
        const int cnt = e - b;
        // most anti-parallel normal pair: min over ordered pairs (a outer, c inner, a != c) of |n_a + n_c|, first minimum wins.
        // k_split_pairs already found, for every member a, its first best partner c BEHIND a (the leaf's first minimal pair has a < c);
        // reduce over a (ties -> smallest a).
        float best = FLT_MAX;
        long long best_pair = LLONG_MAX;
        for (int j0 = b + tid; j0 < e; j0 += 8 * kThreads) {  // eight loads in flight per thread (one at a time: 17 memory latencies for the largest leaves)
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = j0 + u * kThreads < e ? pair_best[j0 + u * kThreads] : ~0ull;  // ~0 = no partner at all
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = v[u] == ~0ull ? FLT_MAX : __uint_as_float((uint32_t)(v[u] >> 32));
                if (d < best) best = d, best_pair = (long long)(j0 + u * kThreads - b) * cnt + (long long)(uint32_t)v[u];  // positions ascend per thread
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ob = __shfl_xor(best, m);
            const long long op = __shfl_xor(best_pair, m);
            if (ob < best || (ob == best && op < best_pair)) best = ob, best_pair = op;
        }
        if (kWaves > 1) {
            __syncthreads();  // previous leaf's readers are done
            if (lane == 0) s_best[wave] = best, s_pair[wave] = best_pair;
            __syncthreads();
            best = s_best[0], best_pair = s_pair[0];
            for (int w = 1; w < kWaves; ++w)
                if (s_best[w] < best || (s_best[w] == best && s_pair[w] < best_pair)) best = s_best[w], best_pair = s_pair[w];
        }
        if (!(best <= 0.5f)) {  // `minDiffFromZero > 0.5f` -> no split (also when there was no pair at all)
            for (int j = b + tid; j < e; j += kThreads) pos_slot_rank[j] = j - b;
            return;  // slot 2l stays as accepted by k_leaf_accept
        }
        const int ia = (int)(best_pair / cnt), ic = (int)(best_pair % cnt);
        const float4 r1 = nsorted[b + ia], r2 = nsorted[b + ic];
        int mn1 = INT_MAX, mx1 = INT_MIN;
        int rank_base1 = 0, rank_base2 = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        auto classify_v = [&](const float4 v, bool in, bool& first, int& id) {   // Gaussians.h:58-75: nearer to the first normal of the pair?
            first = false, id = 0;
            if (in) {
                const float d1 = sqrtf(sum3f((r1.x - v.x) * (r1.x - v.x), (r1.y - v.y) * (r1.y - v.y), (r1.z - v.z) * (r1.z - v.z)));
                const float d2 = sqrtf(sum3f((r2.x - v.x) * (r2.x - v.x), (r2.y - v.y) * (r2.y - v.y), (r2.z - v.z) * (r2.z - v.z)));
                first = d1 < d2;
                id = __float_as_int(v.w);
            }
        };
        auto classify = [&](int j, bool in, bool& first, int& id) {
            first = false, id = 0;
            if (in) {
                const float4 v = nsorted[j];
                const float d1 = sqrtf(sum3f((r1.x - v.x) * (r1.x - v.x), (r1.y - v.y) * (r1.y - v.y), (r1.z - v.z) * (r1.z - v.z)));
                const float d2 = sqrtf(sum3f((r2.x - v.x) * (r2.x - v.x), (r2.y - v.y) * (r2.y - v.y), (r2.z - v.z) * (r2.z - v.z)));
                first = d1 < d2;
                id = __float_as_int(v.w);
            }
        };
        if (kWaves > 1) {
            // Ranks run in position order through the whole leaf.  Every wave takes ONE contiguous run of positions and ranks inside it
            // with counts of its own; the runs' bases need a single exchange, and a second sweep adds them.  (Until round 4 the waves took
            // the positions round by round and exchanged their counts in every round: two workgroup barriers per 1024 members, 35 us for
            // the coarse level's largest leaves.)
            const int run = ((cnt + kWaves - 1) / kWaves + 63) & ~63;
            const int wb = min(e, b + wave * run), we = min(e, wb + run);
            int c1 = 0, c2 = 0;
            for (int jb = wb; jb < we; jb += 4 * 64) {  // four rows of 64 positions: their loads together, their ranks in order
                float4 vv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) vv[u] = jb + 64 * u + lane < we ? nsorted[jb + 64 * u + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = jb + 64 * u + lane;
                    const bool in = j < we;
                    bool first;
                    int id;
                    classify_v(vv[u], in, first, id);
                    const unsigned long long m1 = __ballot(in && first), m2 = __ballot(in && !first);
                    if (in) {
                        if (first) {
                            pos_slot_rank[j] = c1 + __popcll(m1 & below);
                            mn1 = min(mn1, id), mx1 = max(mx1, id);
                        } else {
                            pos_slot_rank[j] = (int32_t)(0x80000000u | (uint32_t)(c2 + __popcll(m2 & below)));
                        }
                    }
                    c1 += __popcll(m1), c2 += __popcll(m2);
                }
            }
            __syncthreads();  // the previous leaf's readers of s_c1 / s_c2 are done
            if (lane == 0) s_c1[wave] = c1, s_c2[wave] = c2;
            __syncthreads();
            int base1 = 0, base2 = 0;
            for (int w = 0; w < kWaves; ++w) {
                if (w == wave) base1 = rank_base1, base2 = rank_base2;
                rank_base1 += s_c1[w], rank_base2 += s_c2[w];
            }
            if ((base1 | base2) != 0)
                for (int jb = wb + lane; jb < we; jb += 8 * 64) {  // the lane that wrote the rank reads it back; eight loads in flight
                    int rr[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) rr[u] = jb + 64 * u < we ? pos_slot_rank[jb + 64 * u] : 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (jb + 64 * u < we) pos_slot_rank[jb + 64 * u] = rr[u] < 0 ? rr[u] + base2 : rr[u] + base1;
                }
        } else {
            for (int j0 = b; j0 < e; j0 += kThreads) {
                const int j = j0 + tid;
                const bool in = j < e;
                bool first;
                int id;
                classify(j, in, first, id);
                const unsigned long long m1 = __ballot(in && first), m2 = __ballot(in && !first);
                if (in) {
                    if (first) {
                        pos_slot_rank[j] = rank_base1 + __popcll(m1 & below);
                        mn1 = min(mn1, id), mx1 = max(mx1, id);
                    } else {
                        pos_slot_rank[j] = (int32_t)(0x80000000u | (uint32_t)(rank_base2 + __popcll(m2 & below)));
                    }
                }
                rank_base1 += __popcll(m1);
                rank_base2 += __popcll(m2);
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            mn1 = min(mn1, __shfl_xor(mn1, m));
            mx1 = max(mx1, __shfl_xor(mx1, m));
        }
        if (kWaves > 1) {
            __syncthreads();
            if (lane == 0) s_mn[wave] = mn1, s_mx[wave] = mx1;
            __syncthreads();
            for (int w = 0; w < kWaves; ++w) mn1 = min(mn1, s_mn[w]), mx1 = max(mx1, s_mx[w]);
        }
        // quirks kept: strict '>' on both sizes and BOTH diversity tests read the first set's ids
        const bool div1 = rank_base1 > 0 && mx1 != mn1;
        if (tid == 0) {
            const int a1 = (rank_base1 > min_pts && div1) ? 1 : 0;
            const int a2 = (rank_base2 > min_pts && div1) ? 1 : 0;
            slot_acc[2 * l] = a1, slot_cnt[2 * l] = a1 ? rank_base1 : 0;
            slot_acc[2 * l + 1] = a2, slot_cnt[2 * l + 1] = a2 ? rank_base2 : 0;
        }
    };
    // leaf l belongs to group l % ngroups (neighbouring leaves are alike: consecutive leaves for one group would hand all leaves of a wall to the
    // same group); thread t of the group looks at the group's t-th leaf of the round
    for (int r0 = 0; group + (int64_t)ngroups * r0 < nl; r0 += kThreads) {
        const int64_t l64 = group + (int64_t)ngroups * (r0 + tid);
        const int l = l64 < nl ? (int)l64 : nl;
        int b = 0, e = 0;
        bool mine = false;
        if (l < nl) {
            const int acc = slot_acc[2 * l];
            b = leaf_start[l], e = leaf_start[l + 1];
            mine = acc != 0 && ((e - b > kSplitBigLeaf) == (kWaves > 1));  // the other kind of group owns the rest
        }
        unsigned long long m = __ballot(mine);
        if (kWaves == 1) {
            while (m != 0ull) {
                const int k = __builtin_ctzll(m);
                m &= m - 1ull;
                process(group + ngroups * (r0 + k), __builtin_amdgcn_readlane(b, k), __builtin_amdgcn_readlane(e, k));
            }
        } else {
            __syncthreads();  // the previous round's readers of s_mask are done
            if (lane == 0) s_mask[wave] = m;
            __syncthreads();
            for (int w = 0; w < kWaves; ++w) {
                unsigned long long mw = s_mask[w];
                while (mw != 0ull) {  // (workgroup-uniform: process() meets at barriers)
                    const int k = __builtin_ctzll(mw);
                    mw &= mw - 1ull;
                    const int lk = group + ngroups * (r0 + 64 * w + k);
                    process(lk, leaf_start[lk], leaf_start[lk + 1]);
                }
            }
        }
    }
}
// One launch for both kinds of group (two until round 4, the second waiting for the first): the first kSplitBigBlocks workgroups take the
// leaves above kSplitBigLeaf members as 1024-thread groups, every wave of the other workgroups is a group of its own for the rest.
constexpr int kSplitBigBlocks = 256, kSplitSmallBlocks = 256;
__global__ __launch_bounds__(1024) void k_leaf_split(const int32_t* __restrict__ leaf_start, const float4* __restrict__ nsorted, const LevelCounts* __restrict__ counts,
                                                     int min_pts, const unsigned long long* __restrict__ pair_best, int32_t* __restrict__ slot_acc,
                                                     int32_t* __restrict__ slot_cnt, int32_t* __restrict__ pos_slot_rank) {
    if (blockIdx.x < kSplitBigBlocks)
        leaf_split_groups<1024>((int)blockIdx.x, kSplitBigBlocks, (int)threadIdx.x, leaf_start, nsorted, counts, min_pts, pair_best, slot_acc, slot_cnt, pos_slot_rank);
    else
        leaf_split_groups<64>((int)(blockIdx.x - kSplitBigBlocks) * 16 + (int)(threadIdx.x >> 6), kSplitSmallBlocks * 16, (int)(threadIdx.x & 63), leaf_start, nsorted,
                              counts, min_pts, pair_best, slot_acc, slot_cnt, pos_slot_rank);
}
// pair_best[n] | counter (own 64-byte slot) | task list.  k_split_tasks emits up to 64 tasks per 64-position wave block, i.e. up to
// 64 * ceil(n / 64) entries when n is not a multiple of 64: the list is sized for that, and the counter no longer sits behind it.
static inline size_t split_task_capacity(int64_t n) { return (size_t)((n + 63) / 64) * 64 + 64; }
// ... | block cones (float4 + float2 per 64 sorted positions)
static inline size_t split_cone_blocks(int64_t n) { return (size_t)((n + 63) / 64) + 1; }
size_t split_scratch_bytes(int64_t n) {
    return (size_t)n * 8 + 64 + split_task_capacity(n) * sizeof(SplitTask) + 64 + split_cone_blocks(n) * 24 + 64 + (size_t)kSplitStripes * 64 + 64;
}
void launch_leaf_split(const int32_t* leaf_incl, const int32_t* leaf_start, const uint32_t* idx_sorted, const int32_t* ring, const float4* nglobal,
                       const LevelCounts* counts, int min_pts, int64_t n, float4* nsorted, unsigned long long* pair_best, int32_t* slot_acc,
                       int32_t* slot_cnt, int32_t* pos_slot_rank, hipStream_t s, unsigned long long* block_stats) {
    // the task counter (own slot) and the task list live behind the n pair_best entries
    int32_t* num_tasks = reinterpret_cast<int32_t*>(pair_best + n);
    SplitTask* tasks = reinterpret_cast<SplitTask*>(reinterpret_cast<char*>(pair_best + n) + 64);
    char* behind = reinterpret_cast<char*>(tasks) + split_task_capacity(n) * sizeof(SplitTask) + 64;
    float4* cone = reinterpret_cast<float4*>(behind);
    float2* norms = reinterpret_cast<float2*>(behind + split_cone_blocks(n) * 16);
    int32_t* tickets = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(behind + split_cone_blocks(n) * 24) + 63) & ~(uintptr_t)63);
    static_assert(kSplitStripes == 256, "k_split_prepare clears one ticket per thread of its first workgroup");
    hipLaunchKernelGGL(k_split_prepare, dim3(grid_for(n, 256)), dim3(256), 0, s, idx_sorted, nglobal, ring, n, nsorted, pair_best, num_tasks, cone, norms, tickets);
    hipLaunchKernelGGL(k_split_tasks, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, s, leaf_incl, leaf_start, slot_acc, n, counts, tasks, num_tasks);
    hipLaunchKernelGGL(k_split_pairs, dim3(4096), dim3(256), 0, s, leaf_incl, leaf_start, slot_acc, nsorted, n, counts, tasks, num_tasks, pair_best, cone, norms,
                       block_stats, tickets);
    hipLaunchKernelGGL(k_leaf_split, dim3(kSplitBigBlocks + kSplitSmallBlocks), dim3(1024), 0, s, leaf_start, nsorted, counts, min_pts, pair_best, slot_acc, slot_cnt,
                       pos_slot_rank);
}

// Exclusive prefix sums of (accepted, accepted member count) over the two slots of every leaf.  The leaf count lives on the
// device, and there are only ~10^4..10^5 leaves, so one 1024-thread workgroup walks them 4096 slots at a time.
__global__ __launch_bounds__(1024) void k_leaf_scan(const int32_t* __restrict__ slot_acc, const int32_t* __restrict__ slot_cnt,
                                                    int32_t* __restrict__ gauss_of_slot, int32_t* __restrict__ memb_of_slot,
                                                    int32_t* __restrict__ pslot_of_slot, LevelCounts* __restrict__ counts) {
    constexpr int kPer = 16;  // slots per thread and round
    __shared__ int s_w[16][3];  // accepted sets, members, members rounded up to 8 (tile slots, see k_tile_rows)
    __shared__ int s_carry[3];
    const int nslots = 2 * counts->num_leaves;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry[0] = 0, s_carry[1] = 0, s_carry[2] = 0;
    __syncthreads();
    for (int base = 0; base < nslots; base += 1024 * kPer) {
        int a[kPer], c[kPer], ta = 0, tc2 = 0, tp = 0;
        const int first = base + kPer * tid;
        if (first + kPer <= nslots) {  // vectorised: 4 x int4 per array
            const int4* pa = reinterpret_cast<const int4*>(slot_acc + first);
            const int4* pc = reinterpret_cast<const int4*>(slot_cnt + first);
#pragma unroll
            for (int k = 0; k < kPer / 4; ++k) {
                const int4 va = pa[k], vc = pc[k];
                a[4 * k] = va.x, a[4 * k + 1] = va.y, a[4 * k + 2] = va.z, a[4 * k + 3] = va.w;
                c[4 * k] = vc.x, c[4 * k + 1] = vc.y, c[4 * k + 2] = vc.z, c[4 * k + 3] = vc.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int i = first + k;
                a[k] = i < nslots ? slot_acc[i] : 0;
                c[k] = i < nslots ? slot_cnt[i] : 0;
            }
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) ta += a[k], tc2 += c[k], tp += pad_slots(c[k]);
        int ia = ta, ic = tc2, ip = tp;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int oa = __shfl_up(ia, d), oc = __shfl_up(ic, d), op = __shfl_up(ip, d);
            if (lane >= d) ia += oa, ic += oc, ip += op;
        }
        if (lane == 63) s_w[wave][0] = ia, s_w[wave][1] = ic, s_w[wave][2] = ip;
        __syncthreads();
        int ra = s_carry[0] + ia - ta, rc = s_carry[1] + ic - tc2, rp = s_carry[2] + ip - tp;
        for (int w = 0; w < wave; ++w) ra += s_w[w][0], rc += s_w[w][1], rp += s_w[w][2];
        int oa[kPer], oc[kPer], op[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            oa[k] = ra, oc[k] = rc, op[k] = rp;
            ra += a[k], rc += c[k], rp += pad_slots(c[k]);
        }
        if (first + kPer <= nslots) {  // 16-byte stores: 4x fewer (lane-strided) store instructions
            int4* qa = reinterpret_cast<int4*>(gauss_of_slot + first);
            int4* qc = reinterpret_cast<int4*>(memb_of_slot + first);
            int4* qp = reinterpret_cast<int4*>(pslot_of_slot + first);
#pragma unroll
            for (int k = 0; k < kPer / 4; ++k) {
                qa[k] = make_int4(oa[4 * k], oa[4 * k + 1], oa[4 * k + 2], oa[4 * k + 3]);
                qc[k] = make_int4(oc[4 * k], oc[4 * k + 1], oc[4 * k + 2], oc[4 * k + 3]);
                qp[k] = make_int4(op[4 * k], op[4 * k + 1], op[4 * k + 2], op[4 * k + 3]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < kPer; ++k)
                if (first + k < nslots) gauss_of_slot[first + k] = oa[k], memb_of_slot[first + k] = oc[k], pslot_of_slot[first + k] = op[k];
        }
        __syncthreads();
        if (tid == 1023) s_carry[0] = ra, s_carry[1] = rc, s_carry[2] = rp;
        __syncthreads();
    }
    if (tid == 0) counts->num_gauss = s_carry[0], counts->num_memb = s_carry[1], counts->pad = s_carry[2];  // pad: tile slots of the level
}
void launch_leaf_scan(const int32_t* slot_acc, const int32_t* slot_cnt, int32_t* gauss_of_slot, int32_t* memb_of_slot, int32_t* pslot_of_slot, LevelCounts* counts,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_leaf_scan, dim3(1), dim3(1024), 0, s, slot_acc, slot_cnt, gauss_of_slot, memb_of_slot, pslot_of_slot, counts);
}

// The same three slot scans as ONE multi-workgroup pass (the single-workgroup kernel above walks the 3 x 10^5 leaves of a window's fine
// level in 24 us).  Tiles of 4096 leaves are handed out by an atomic ticket; a tile publishes its three totals (accepted sets, members,
// members rounded up to 8) and looks back over the earlier tiles 64 at a time, exactly like k_leaf_segments: every published word
// carries the call's epoch, partial totals and inclusive prefixes live in separate words (three values cannot change state atomically
// together), nothing is cleared between calls.  (Computing the leaf test of k_leaf_accept in here as well was measured 15 us SLOWER:
// eight leaves per thread walk their members one after the other, while k_leaf_accept spreads the walks over 131 072 threads.)
constexpr int kFinThreads = 512, kFinItems = 8, kFinTile = kFinThreads * kFinItems;  // leaves per tile
constexpr int kFinWords = 6;                                                          // per tile: partial a, c, p | prefix a, c, p
__global__ __launch_bounds__(kFinThreads) void k_leaf_finalize(const int32_t* __restrict__ slot_acc, const int32_t* __restrict__ slot_cnt,
                                                                int32_t* __restrict__ gauss_of_slot,
                                                                int32_t* __restrict__ memb_of_slot, int32_t* __restrict__ pslot_of_slot,
                                                                LevelCounts* __restrict__ counts, unsigned long long* __restrict__ state /* [0]: ticket */,
                                                                uint32_t epoch, uint32_t ticket_base) {
    __shared__ uint32_t s_tile;
    __shared__ int s_wave_total[kFinThreads / 64][3];
    __shared__ int s_excl[3];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) s_tile = atomicAdd(reinterpret_cast<unsigned int*>(state), 1u) - ticket_base;
    __syncthreads();
    const uint32_t tile = s_tile;
    const int nl = counts->num_leaves;
    const int base = (int)tile * kFinTile;
    if (base >= nl && !(tile == 0)) return;  // surplus tile (the grid is sized for one leaf per point); tile 0 always reports the totals
    int a0[kFinItems], c0[kFinItems], a1[kFinItems], c1[kFinItems];
#pragma unroll
    for (int k = 0; k < kFinItems; ++k) {
        const int l = base + (wave * kFinItems + k) * 64 + lane;
        int2 va = make_int2(0, 0), vc = make_int2(0, 0);
        if (l < nl) va = reinterpret_cast<const int2*>(slot_acc)[l], vc = reinterpret_cast<const int2*>(slot_cnt)[l];
        a0[k] = va.x, a1[k] = va.y, c0[k] = vc.x, c1[k] = vc.y;
    }
    // inclusive scans over the leaves of the wave's rows (a leaf = its two slots)
    int ia[kFinItems], ic[kFinItems], ip[kFinItems];
    int ra = 0, rc = 0, rp = 0;
#pragma unroll
    for (int k = 0; k < kFinItems; ++k) {
        const int sa = wave_incl_scan_dpp(a0[k] + a1[k]), sc = wave_incl_scan_dpp(c0[k] + c1[k]), sp = wave_incl_scan_dpp(pad_slots(c0[k]) + pad_slots(c1[k]));
        ia[k] = ra + sa, ic[k] = rc + sc, ip[k] = rp + sp;
        ra += __builtin_amdgcn_readlane(sa, 63), rc += __builtin_amdgcn_readlane(sc, 63), rp += __builtin_amdgcn_readlane(sp, 63);
    }
    if (lane == 0) s_wave_total[wave][0] = ra, s_wave_total[wave][1] = rc, s_wave_total[wave][2] = rp;
    __syncthreads();
    int before[3] = {0, 0, 0}, total[3] = {0, 0, 0};
#pragma unroll
    for (int w = 0; w < kFinThreads / 64; ++w)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int t = s_wave_total[w][q];
            if (w < wave) before[q] += t;
            total[q] += t;
        }
    if (wave == 0) {
        const unsigned long long tag = (unsigned long long)epoch << 32;
        unsigned long long* st = state + 1;
        int excl[3] = {0, 0, 0};
        if (tile != 0) {
            if (lane < 3) __hip_atomic_store(st + (size_t)tile * kFinWords + lane, tag | (unsigned)total[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t t = (int64_t)tile - 1;
            while (true) {
                const int64_t mine = t - lane;
                unsigned long long v[kFinWords];
#pragma unroll
                for (int q = 0; q < kFinWords; ++q)
                    v[q] = mine >= 0 ? __hip_atomic_load(st + (size_t)mine * kFinWords + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tag;
                const bool has_part = (uint32_t)(v[0] >> 32) == epoch && (uint32_t)(v[1] >> 32) == epoch && (uint32_t)(v[2] >> 32) == epoch;
                const bool has_pref = (uint32_t)(v[3] >> 32) == epoch && (uint32_t)(v[4] >> 32) == epoch && (uint32_t)(v[5] >> 32) == epoch;
                const unsigned long long ready = __ballot(has_part || has_pref), prefix = __ballot(has_pref);
                const int first_missing = ready == ~0ull ? 64 : __builtin_ctzll(~ready);
                const int first_prefix = prefix == 0ull ? 64 : __builtin_ctzll(prefix);
                const bool done = first_prefix < first_missing;
                const int stop = done ? first_prefix + 1 : first_missing;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    int part = lane < stop ? (int)(uint32_t)(has_pref ? v[3 + q] : v[q]) : 0;
                    part = wave_incl_scan_dpp(part);
                    excl[q] += __builtin_amdgcn_readlane(part, 63);
                }
                if (done) break;
                t -= stop;
            }
        }
        if (lane < 3)
            __hip_atomic_store(st + (size_t)tile * kFinWords + 3 + lane, tag | (unsigned)(excl[lane] + total[lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) s_excl[0] = excl[0], s_excl[1] = excl[1], s_excl[2] = excl[2];
    }
    __syncthreads();
    const int oa = s_excl[0] + before[0], oc = s_excl[1] + before[1], op = s_excl[2] + before[2];
#pragma unroll
    for (int k = 0; k < kFinItems; ++k) {
        const int l = base + (wave * kFinItems + k) * 64 + lane;
        if (l < nl) {
            // exclusive prefix of slot 2l = everything before the leaf; slot 2l + 1 additionally has slot 2l in front of it
            const int ea = oa + ia[k] - (a0[k] + a1[k]), ec = oc + ic[k] - (c0[k] + c1[k]), ep = op + ip[k] - (pad_slots(c0[k]) + pad_slots(c1[k]));
            reinterpret_cast<int2*>(gauss_of_slot)[l] = make_int2(ea, ea + a0[k]);
            reinterpret_cast<int2*>(memb_of_slot)[l] = make_int2(ec, ec + c0[k]);
            reinterpret_cast<int2*>(pslot_of_slot)[l] = make_int2(ep, ep + pad_slots(c0[k]));
        }
    }
    // the tile that holds the last leaf reports the totals of the level (tile 0 when there is no leaf at all)
    if (tid == 0 && (nl <= 0 ? tile == 0 : (base < nl && nl <= base + kFinTile)))
        counts->num_gauss = s_excl[0] + total[0], counts->num_memb = s_excl[1] + total[1], counts->pad = s_excl[2] + total[2];
}
int leaf_finalize_tiles(int64_t n) { return (int)std::max<int64_t>(1, (n + kFinTile - 1) / kFinTile); }
size_t leaf_finalize_state_bytes(int64_t n) { return 8 * (size_t)(1 + kFinWords * leaf_finalize_tiles(n)); }
void launch_leaf_finalize(const int32_t* slot_acc, const int32_t* slot_cnt, int64_t n, int32_t* gauss_of_slot, int32_t* memb_of_slot, int32_t* pslot_of_slot,
                          LevelCounts* counts, unsigned long long* state, uint32_t epoch, uint32_t ticket_base, hipStream_t s) {
    hipLaunchKernelGGL(k_leaf_finalize, dim3((unsigned)leaf_finalize_tiles(n)), dim3(kFinThreads), 0, s, slot_acc, slot_cnt, gauss_of_slot, memb_of_slot, pslot_of_slot,
                       counts, state, epoch, ticket_base);
}


// members of accepted sets, physically regrouped in Gaussian order (leaf DFS order, ascending point index inside a
// set): the correspondence kernel then streams contiguous float4s instead of gathering through index lists.
template <typename KeyT>
__global__ __launch_bounds__(256) void k_gather_members(const int32_t* __restrict__ leaf_incl, const int32_t* __restrict__ leaf_start,
                                                        const uint32_t* __restrict__ idx_sorted, const KeyT* __restrict__ code,
                                                        const LatticeTable* __restrict__ table, const int32_t* __restrict__ slot_acc,
                                                        const int32_t* __restrict__ gauss_of_slot, const int32_t* __restrict__ memb_of_slot,
                                                        const int32_t* __restrict__ pos_slot_rank, const float4* __restrict__ local,
                                                        const int32_t* __restrict__ slot_cnt, const GaussCounts* __restrict__ counts, int level,
                                                        int64_t n, float4* __restrict__ memb_local, int32_t* __restrict__ memb_idx,
                                                        int32_t* __restrict__ memb_g, int32_t* __restrict__ seg_off,
                                                        const int32_t* __restrict__ pslot_of_slot, int32_t* __restrict__ pad_off) {
    const KeyT invalid = (KeyT)lattice_invalid_code(*table);
    const int gbase = level == 0 ? 0 : counts->level[0].num_gauss;
    const int mbase = level == 0 ? 0 : counts->level[0].num_memb;
    const int pbase = level == 0 ? 0 : counts->level[0].pad;  // tile-slot offsets continue behind level 0 like the member offsets
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid0 == 0) {
        seg_off[gbase + counts->level[level].num_gauss] = mbase + counts->level[level].num_memb;
        pad_off[gbase + counts->level[level].num_gauss] = pbase + counts->level[level].pad;
    }
    for (int64_t i = tid0; i < n; i += stride) {
        if (code[i] == invalid) continue;
        const int l = leaf_incl[i] - 1;
        int set = 0, rank = (int)(i - leaf_start[l]);
        if (pos_slot_rank != nullptr && slot_acc[2 * l] + slot_acc[2 * l + 1] > 0) {
            // leaves that went through k_leaf_split carry explicit (set, rank); untouched leaves are rejected anyway
            const int32_t v = pos_slot_rank[i];
            set = v < 0 ? 1 : 0;
            rank = v & 0x7fffffff;
        }
        const int slot = 2 * l + set;
        if (!slot_acc[slot]) continue;
        const int g = gbase + gauss_of_slot[slot];
        const int dst = mbase + memb_of_slot[slot] + rank;
        const uint32_t pi = idx_sorted[i];
        memb_local[dst] = local[pi];
        memb_idx[dst] = (int32_t)pi;
        memb_g[dst] = (int32_t)((uint32_t)g | (rank == slot_cnt[slot] - 1 ? 0x80000000u : 0u));  // Gaussian id, bit 31: last member
        if (rank == 0) seg_off[g] = dst, pad_off[g] = pbase + pslot_of_slot[slot];
    }
}
void launch_gather_members(const int32_t* leaf_of_pos, const int32_t* leaf_start, const uint32_t* idx_sorted, const void* code_sorted, bool key32,
                           const LatticeTable* table, const int32_t* slot_acc, const int32_t* gauss_of_slot, const int32_t* memb_of_slot,
                           const int32_t* pos_slot_rank, const float4* local, const int32_t* slot_cnt, const GaussCounts* counts, int level, int64_t n,
                           float4* memb_local, int32_t* memb_idx, int32_t* memb_g, int32_t* seg_off, const int32_t* pslot_of_slot, int32_t* pad_off,
                           hipStream_t s) {
    if (n <= 0) return;
    if (key32)
        hipLaunchKernelGGL(k_gather_members<uint32_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, leaf_of_pos, leaf_start, idx_sorted,
                           (const uint32_t*)code_sorted, table, slot_acc, gauss_of_slot, memb_of_slot, pos_slot_rank, local, slot_cnt, counts, level, n,
                           memb_local, memb_idx, memb_g, seg_off, pslot_of_slot, pad_off);
    else
        hipLaunchKernelGGL(k_gather_members<uint64_t>, dim3(grid_for(n, 256)), dim3(256), 0, s, leaf_of_pos, leaf_start, idx_sorted,
                           (const uint64_t*)code_sorted, table, slot_acc, gauss_of_slot, memb_of_slot, pos_slot_rank, local, slot_cnt, counts, level, n,
                           memb_local, memb_idx, memb_g, seg_off, pslot_of_slot, pad_off);
}

// ------------------------------------------------------------------------------------------------------------
// K3 — Gaussian fit: covariance (double accumulation, rounded once), eigenvalue clamp, information matrix
// ------------------------------------------------------------------------------------------------------------
__device__ void inverse3_f(const float m[3][3], float inv[3][3]);

// ---- EigenSolver<Matrix3f> as Eigen 3.4.0 runs it on the reference's covariances (Gaussians.h:184-188) ----------------------------------
// One lane = one Gaussian.  The statements are those of Eigen's RealSchur / HessenbergDecomposition / Householder / Jacobi / EigenSolver
// sources, hand-specialised for a 3 x 3 matrix (indices are compile-time constants, the matrices live in registers):
//   scale by the largest |coefficient|; ONE Householder reflector brings the matrix to Hessenberg form (the second reflector of
//   HessenbergDecomposition::_compute acts on a vector of one coefficient: tau = 0, multiplications by 1.0f); U = the reflector applied to I;
//   RealSchur::computeFromHessenberg: while the active window is the full matrix (il = 0, iu = 2) Francis double-shift steps -- shift from the
//   trailing 2 x 2 block (Wilkinson's / MATLAB's exceptional shifts at the 10th / 30th step), a reflector of the 3-vector, a reflector of the
//   2-vector below it --, then a trailing row deflates (il = 2) or the trailing 2 x 2 block splits off by a Givens rotation (il = 1), and the
//   remaining 2 x 2 / 1 x 1 part does the same; T *= scale; eigenvalues off T's diagonal in that order; EigenSolver::doComputeEigenvectors:
//   back substitution on T, then eivec = U * T column by column (the 3-term column sequentially); eigenvectors(): columns divided by their norm.
// Every multiply, add, divide and sqrt rounds once (no FMA: -ffp-contract=off; IEEE division and sqrt), reflector updates multiply
// (tau * essential_i) * tmp_j from the left and (tau * tmp_i) * essential_j from the right as Eigen's expressions associate.  A complex pair
// (a trailing block of pure rounding noise) follows EigenSolver's complex branch with libgcc 9's float Smith division; the real parts of the
// two eigenvectors then coincide, V is singular and the Gaussian's information matrix is not finite -- in the reference as well.
// std::max / numext::maxi: (a < b) ? b : a -- a NaN in the first place stays, one in the second is dropped
__device__ __forceinline__ float eig3_max(float a, float b) { return a < b ? b : a; }
struct Eig3 {
    float t[3][3], u[3][3];
};
// Householder.h makeHouseholder for a 3-vector / 2-vector: essential part, tau, beta
__device__ __forceinline__ void eig3_householder3(float c0, float v1, float v2, float& e0, float& e1, float& tau, float& beta) {
    const float tail = v1 * v1 + v2 * v2;
    if (tail <= FLT_MIN) {
        tau = 0.0f, beta = c0, e0 = 0.0f, e1 = 0.0f;
    } else {
        float b = sqrtf(c0 * c0 + tail);
        if (c0 >= 0.0f) b = -b;
        const float d = c0 - b;
        e0 = v1 / d, e1 = v2 / d;
        tau = (b - c0) / b, beta = b;
    }
}
__device__ __forceinline__ void eig3_householder2(float c0, float v1, float& e, float& tau, float& beta) {
    const float tail = v1 * v1;
    if (tail <= FLT_MIN) {
        tau = 0.0f, beta = c0, e = 0.0f;
    } else {
        float b = sqrtf(c0 * c0 + tail);
        if (c0 >= 0.0f) b = -b;
        e = v1 / (c0 - b);
        tau = (b - c0) / b, beta = b;
    }
}
// reflector (1, e) on rows R, R + 1 of m, columns C0 .. 2 (applyHouseholderOnTheLeft of a 2-row block)
template <int R, int C0>
__device__ __forceinline__ void eig3_left2(float (&m)[3][3], float e, float tau) {
    if (tau == 0.0f) return;
    const float te = tau * e;
#pragma unroll
    for (int j = C0; j < 3; ++j) {
        float tmp = e * m[R + 1][j];
        tmp = tmp + m[R][j];
        m[R][j] = m[R][j] - tau * tmp;
        m[R + 1][j] = m[R + 1][j] - te * tmp;
    }
}
// reflector (1, e) on columns C, C + 1 of m, all three rows (applyHouseholderOnTheRight of a 2-column block)
template <int C>
__device__ __forceinline__ void eig3_right2(float (&m)[3][3], float e, float tau) {
    if (tau == 0.0f) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float tmp = m[i][C + 1] * e;
        tmp = tmp + m[i][C];
        const float tt = tau * tmp;
        m[i][C] = m[i][C] - tt;
        m[i][C + 1] = m[i][C + 1] - tt * e;
    }
}
// reflector (1, e0, e1) on the whole matrix from the left / from the right
__device__ __forceinline__ void eig3_left3(float (&m)[3][3], float e0, float e1, float tau) {
    if (tau == 0.0f) return;
    const float te0 = tau * e0, te1 = tau * e1;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float tmp = e0 * m[1][j] + e1 * m[2][j];
        tmp = tmp + m[0][j];
        m[0][j] = m[0][j] - tau * tmp;
        m[1][j] = m[1][j] - te0 * tmp;
        m[2][j] = m[2][j] - te1 * tmp;
    }
}
__device__ __forceinline__ void eig3_right3(float (&m)[3][3], float e0, float e1, float tau) {
    if (tau == 0.0f) return;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float tmp = m[i][1] * e0 + m[i][2] * e1;
        tmp = tmp + m[i][0];
        const float tt = tau * tmp;
        m[i][0] = m[i][0] - tt;
        m[i][1] = m[i][1] - tt * e0;
        m[i][2] = m[i][2] - tt * e1;
    }
}
// RealSchur::splitOffTwoRows(IU): the block of rows / columns IU - 1, IU decouples
template <int IU>
__device__ __forceinline__ void eig3_split(Eig3& s, float exshift) {
    constexpr int A = IU - 1, B = IU;
    const float p = 0.5f * (s.t[A][A] - s.t[B][B]);
    const float q = p * p + s.t[B][A] * s.t[A][B];
    s.t[B][B] = s.t[B][B] + exshift;
    s.t[A][A] = s.t[A][A] + exshift;
    if (q >= 0.0f) {
        const float z = sqrtf(fabsf(q));
        // JacobiRotation::makeGivens(p +- z, T(iu, iu - 1))
        const float gp = p >= 0.0f ? p + z : p - z, gq = s.t[B][A];
        float c, sn;
        if (gq == 0.0f) {
            c = gp < 0.0f ? -1.0f : 1.0f, sn = 0.0f;
        } else if (gp == 0.0f) {
            c = 0.0f, sn = gq < 0.0f ? 1.0f : -1.0f;
        } else if (fabsf(gp) > fabsf(gq)) {
            const float t = gq / gp;
            float w = sqrtf(1.0f + t * t);
            if (gp < 0.0f) w = -w;
            c = 1.0f / w;
            sn = -t * c;
        } else {
            const float t = gp / gq;
            float w = sqrtf(1.0f + t * t);
            if (gq < 0.0f) w = -w;
            sn = -1.0f / w;
            c = -t * sn;
        }
        // both applications use the rotation (c, -s): x <- c x + (-s) y, y <- s x + c y; the identity rotation is skipped
        const float ms = -sn;
        if (!(c == 1.0f && ms == 0.0f)) {
#pragma unroll
            for (int j = A; j < 3; ++j) {  // rightCols(size - iu + 1).applyOnTheLeft(iu - 1, iu, rot.adjoint())
                const float x = s.t[A][j], y = s.t[B][j];
                s.t[A][j] = c * x + ms * y;
                s.t[B][j] = -ms * x + c * y;
            }
#pragma unroll
            for (int i = 0; i <= B; ++i) {  // topRows(iu + 1).applyOnTheRight(iu - 1, iu, rot)
                const float x = s.t[i][A], y = s.t[i][B];
                s.t[i][A] = c * x + ms * y;
                s.t[i][B] = -ms * x + c * y;
            }
        }
        s.t[B][A] = 0.0f;
        if (!(c == 1.0f && ms == 0.0f)) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {  // m_matU.applyOnTheRight(iu - 1, iu, rot)
                const float x = s.u[i][A], y = s.u[i][B];
                s.u[i][A] = c * x + ms * y;
                s.u[i][B] = -ms * x + c * y;
            }
        }
    }
    if (IU > 1) s.t[1][0] = 0.0f;
}
// std::complex<float> division, libgcc 9's __divsc3 (Smith) without its NaN-recovery tail
__device__ __forceinline__ void eig3_cdiv(float a, float b, float c, float d, float& x, float& y) {
    if (fabsf(c) < fabsf(d)) {
        const float ratio = c / d;
        const float denom = (c * ratio) + d;
        x = ((a * ratio) + b) / denom;
        y = ((b * ratio) - a) / denom;
    } else {
        const float ratio = d / c;
        const float denom = (d * ratio) + c;
        x = ((b * ratio) + a) / denom;
        y = (b - (a * ratio)) / denom;
    }
}
// returns 0 (Success), 1 (NumericalIssue), 2 (NoConvergence); lam = eigenvalues().real(), v = eigenvectors().real(); *iters = Francis steps
__device__ int eigensolver3f(const float c[3][3], float v[3][3], float lam[3], int* iters) {
    Eig3 s;
    float scale = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) scale = eig3_max(scale, fabsf(c[i][j]));
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s.u[i][j] = i == j ? 1.0f : 0.0f;
    if (iters) *iters = 0;
    if (scale < FLT_MIN) {  // T = 0, U = I: eigenvalues 0, doComputeEigenvectors returns at once, the columns of I are normalised by 1
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lam[i] = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) v[i][j] = s.u[i][j] / 1.0f;
        }
        return 0;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) s.t[i][j] = c[i][j] / scale;
    {   // HessenbergDecomposition::_compute, i = 0 (i = 1 multiplies by 1.0f) and matrixQ().evalTo
        float e, tau, beta;
        eig3_householder2(s.t[1][0], s.t[2][0], e, tau, beta);
        s.t[1][0] = beta;
        eig3_left2<1, 1>(s.t, e, tau);
        eig3_right2<1>(s.t, e, tau);
        s.t[2][0] = 0.0f;  // matrixH(): below the subdiagonal cleared (the essential part lived there)
        eig3_left2<1, 1>(s.u, e, tau);
    }
    // RealSchur::computeFromHessenberg
    float norm = 0.0f;
    norm = norm + (fabsf(s.t[0][0]) + fabsf(s.t[1][0]));
    norm = norm + ((fabsf(s.t[0][1]) + fabsf(s.t[1][1])) + fabsf(s.t[2][1]));
    norm = norm + ((fabsf(s.t[0][2]) + fabsf(s.t[1][2])) + fabsf(s.t[2][2]));
    const float caz = eig3_max(norm * (FLT_EPSILON * FLT_EPSILON), FLT_MIN);
    float exshift = 0.0f;
    int total = 0;
    if (norm != 0.0f) {
        int iter = 0, il;
        for (;;) {  // active window = rows 0 .. 2
            {       // findSmallSubdiagEntry(2)
                float sm = eig3_max((fabsf(s.t[1][1]) + fabsf(s.t[2][2])) * FLT_EPSILON, caz);
                if (fabsf(s.t[2][1]) <= sm)
                    il = 2;
                else {
                    sm = eig3_max((fabsf(s.t[0][0]) + fabsf(s.t[1][1])) * FLT_EPSILON, caz);
                    il = fabsf(s.t[1][0]) <= sm ? 1 : 0;
                }
            }
            if (il != 0) break;
            // computeShift(2, iter)
            float s0 = s.t[2][2], s1 = s.t[1][1], s2 = s.t[2][1] * s.t[1][2];
            if (iter == 10) {
                exshift = exshift + s0;
                s.t[0][0] = s.t[0][0] - s0, s.t[1][1] = s.t[1][1] - s0, s.t[2][2] = s.t[2][2] - s0;
                const float w = fabsf(s.t[2][1]) + fabsf(s.t[1][0]);
                s0 = 0.75f * w, s1 = 0.75f * w, s2 = -0.4375f * w * w;
            }
            if (iter == 30) {
                float w = (s1 - s0) / 2.0f;
                w = w * w + s2;
                if (w > 0.0f) {
                    w = sqrtf(w);
                    if (s1 < s0) w = -w;
                    w = w + (s1 - s0) / 2.0f;
                    w = s0 - s2 / w;
                    exshift = exshift + w;
                    s.t[0][0] = s.t[0][0] - w, s.t[1][1] = s.t[1][1] - w, s.t[2][2] = s.t[2][2] - w;
                    s0 = s1 = s2 = 0.964f;
                }
            }
            iter = iter + 1;
            total = total + 1;
            if (total > 120) break;
            // initFrancisQRStep: im = il = 0
            float e0, e1, tau, beta;
            {
                const float Tmm = s.t[0][0];
                const float r = s0 - Tmm, q = s1 - Tmm;
                const float v0 = (r * q - s2) / s.t[1][0] + s.t[0][1];
                const float v1 = s.t[1][1] - Tmm - r - q;
                const float v2 = s.t[2][1];
                eig3_householder3(v0, v1, v2, e0, e1, tau, beta);
            }
            // performFrancisQRStep: k = 0, then the 2-vector below
            if (beta != 0.0f) {
                eig3_left3(s.t, e0, e1, tau);
                eig3_right3(s.t, e0, e1, tau);
                eig3_right3(s.u, e0, e1, tau);
            }
            float e;
            eig3_householder2(s.t[1][0], s.t[2][0], e, tau, beta);
            if (beta != 0.0f) {
                s.t[1][0] = beta;
                eig3_left2<1, 1>(s.t, e, tau);
                eig3_right2<1>(s.t, e, tau);
                eig3_right2<1>(s.u, e, tau);
            }
            s.t[2][0] = 0.0f;
        }
        if (total > 120) {
            if (iters) *iters = total;
            return 2;
        }
        if (il == 2) {  // one root found: the rest is the leading 2 x 2 block
            s.t[2][2] = s.t[2][2] + exshift;
            s.t[2][1] = 0.0f;
            const float sm = eig3_max((fabsf(s.t[0][0]) + fabsf(s.t[1][1])) * FLT_EPSILON, caz);
            if (fabsf(s.t[1][0]) <= sm) {
                s.t[1][1] = s.t[1][1] + exshift;
                s.t[1][0] = 0.0f;
                s.t[0][0] = s.t[0][0] + exshift;
            } else {
                eig3_split<1>(s, exshift);
            }
        } else {  // two roots found: rows 1, 2 split off, row 0 is a root
            eig3_split<2>(s, exshift);
            s.t[0][0] = s.t[0][0] + exshift;
        }
    }
    if (iters) *iters = total;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) s.t[i][j] = s.t[i][j] * scale;
    // EigenSolver::compute: eigenvalues from T.  kind 0: three real; 1: a pair in rows 0, 1; 2: a pair in rows 1, 2
    float re[3], im[3] = {0.0f, 0.0f, 0.0f};
    int kind = 0;
    auto pair_values = [&](float taa, float tab, float tba, float tbb, float& r, float& z) {
        const float p = 0.5f * (taa - tbb);
        float t0 = tba, t1 = tab;
        const float maxval = eig3_max(fabsf(p), eig3_max(fabsf(t0), fabsf(t1)));
        t0 = t0 / maxval, t1 = t1 / maxval;
        const float p0 = p / maxval;
        z = maxval * sqrtf(fabsf(p0 * p0 + t0 * t1));
        r = tbb + p;
    };
    if (s.t[1][0] != 0.0f) {
        kind = 1;
        float r, z;
        pair_values(s.t[0][0], s.t[0][1], s.t[1][0], s.t[1][1], r, z);
        re[0] = r, im[0] = z, re[1] = r, im[1] = -z, re[2] = s.t[2][2];
    } else if (s.t[2][1] != 0.0f) {
        kind = 2;
        float r, z;
        pair_values(s.t[1][1], s.t[1][2], s.t[2][1], s.t[2][2], r, z);
        re[0] = s.t[0][0], re[1] = r, im[1] = z, re[2] = r, im[2] = -z;
    } else {
        re[0] = s.t[0][0], re[1] = s.t[1][1], re[2] = s.t[2][2];
    }
    {
        bool fin = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) fin &= isfinite(re[k]) && isfinite(im[k]);
        if (!fin) return 1;
    }
    // doComputeEigenvectors
    float nrm = 0.0f;
    nrm = nrm + ((fabsf(s.t[0][0]) + fabsf(s.t[0][1])) + fabsf(s.t[0][2]));
    nrm = nrm + ((fabsf(s.t[1][0]) + fabsf(s.t[1][1])) + fabsf(s.t[1][2]));
    nrm = nrm + (fabsf(s.t[2][1]) + fabsf(s.t[2][2]));
    float ev[3][3];  // m_eivec
    if (nrm == 0.0f) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) ev[i][j] = s.u[i][j];
    } else {
        const float eps = FLT_EPSILON;
        if (kind == 0) {
            {   // n = 2
                const float p = re[2];
                s.t[2][2] = 1.0f;
                float w = s.t[1][1] - p;
                float r = s.t[1][2] * s.t[2][2];
                s.t[1][2] = w != 0.0f ? -r / w : -r / (eps * nrm);
                float t = fabsf(s.t[1][2]);
                if ((eps * t) * t > 1.0f) s.t[1][2] = s.t[1][2] / t, s.t[2][2] = s.t[2][2] / t;
                w = s.t[0][0] - p;
                r = s.t[0][1] * s.t[1][2] + s.t[0][2] * s.t[2][2];
                s.t[0][2] = w != 0.0f ? -r / w : -r / (eps * nrm);
                t = fabsf(s.t[0][2]);
                if ((eps * t) * t > 1.0f) s.t[0][2] = s.t[0][2] / t, s.t[1][2] = s.t[1][2] / t, s.t[2][2] = s.t[2][2] / t;
            }
            {   // n = 1
                const float p = re[1];
                s.t[1][1] = 1.0f;
                const float w = s.t[0][0] - p;
                const float r = s.t[0][1] * s.t[1][1];
                s.t[0][1] = w != 0.0f ? -r / w : -r / (eps * nrm);
                const float t = fabsf(s.t[0][1]);
                if ((eps * t) * t > 1.0f) s.t[0][1] = s.t[0][1] / t, s.t[1][1] = s.t[1][1] / t, s.t[2][1] = s.t[2][1] / t;
            }
            s.t[0][0] = 1.0f;  // n = 0
        } else if (kind == 2) {
            {   // n = 2: the complex vector of the pair in rows 1, 2
                const float p = re[2], q = im[2];
                if (fabsf(s.t[2][1]) > fabsf(s.t[1][2])) {
                    s.t[1][1] = q / s.t[2][1];
                    s.t[1][2] = -(s.t[2][2] - p) / s.t[2][1];
                } else {
                    float cr, ci;
                    eig3_cdiv(0.0f, -s.t[1][2], s.t[1][1] - p, q, cr, ci);
                    s.t[1][1] = cr, s.t[1][2] = ci;
                }
                s.t[2][1] = 0.0f;
                s.t[2][2] = 1.0f;
                const float ra = s.t[0][1] * s.t[1][1] + s.t[0][2] * s.t[2][1];
                const float sa = s.t[0][1] * s.t[1][2] + s.t[0][2] * s.t[2][2];
                const float w = s.t[0][0] - p;
                float cr, ci;
                eig3_cdiv(-ra, -sa, w, q, cr, ci);
                s.t[0][1] = cr, s.t[0][2] = ci;
                const float t = eig3_max(fabsf(s.t[0][1]), fabsf(s.t[0][2]));
                if ((eps * t) * t > 1.0f) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) s.t[k][1] = s.t[k][1] / t, s.t[k][2] = s.t[k][2] / t;
                }
            }
            s.t[0][0] = 1.0f;  // n = 0
        } else {
            {   // n = 2: a real vector above the pair in rows 0, 1
                const float p = re[2];
                s.t[2][2] = 1.0f;
                const float lastw = s.t[1][1] - p;
                const float lastr = s.t[1][2] * s.t[2][2];
                const float w = s.t[0][0] - p;
                const float r = s.t[0][2] * s.t[2][2];
                const float x = s.t[0][1], y = s.t[1][0];
                const float denom = (re[0] - p) * (re[0] - p) + im[0] * im[0];
                const float t = (x * lastr - lastw * r) / denom;
                s.t[0][2] = t;
                if (fabsf(x) > fabsf(lastw))
                    s.t[1][2] = (-r - w * t) / x;
                else
                    s.t[1][2] = (-lastr - y * t) / lastw;
                const float t2 = fabsf(s.t[0][2]);
                if ((eps * t2) * t2 > 1.0f) s.t[0][2] = s.t[0][2] / t2, s.t[1][2] = s.t[1][2] / t2, s.t[2][2] = s.t[2][2] / t2;
            }
            {   // n = 1: the complex vector of the pair
                const float p = re[1], q = im[1];
                if (fabsf(s.t[1][0]) > fabsf(s.t[0][1])) {
                    s.t[0][0] = q / s.t[1][0];
                    s.t[0][1] = -(s.t[1][1] - p) / s.t[1][0];
                } else {
                    float cr, ci;
                    eig3_cdiv(0.0f, -s.t[0][1], s.t[0][0] - p, q, cr, ci);
                    s.t[0][0] = cr, s.t[0][1] = ci;
                }
                s.t[1][0] = 0.0f;
                s.t[1][1] = 1.0f;
            }
        }
        // back transformation: column j of U * T's leading (j + 1) x (j + 1) part, the products summed from the first
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ev[i][2] = (s.u[i][0] * s.t[0][2] + s.u[i][1] * s.t[1][2]) + s.u[i][2] * s.t[2][2];
            ev[i][1] = s.u[i][0] * s.t[0][1] + s.u[i][1] * s.t[1][1];
            ev[i][0] = s.u[i][0] * s.t[0][0];
        }
    }
    // eigenvectors(): columns normalised (abs2 of a complex coefficient = re * re + im * im, summed x0 + (x1 + x2)); .real()
    const float precision = 2.0f * FLT_EPSILON;
#pragma unroll
    for (int j = 0; j < 3; ++j) lam[j] = re[j];
    bool pair_here[3] = {false, false, false};  // column j starts a pair that eigenvectors() treats as one
    if (kind == 1) pair_here[0] = !(fabsf(im[0]) <= fabsf(re[0]) * precision);
    if (kind == 2) pair_here[1] = !(fabsf(im[1]) <= fabsf(re[1]) * precision);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const bool second = j > 0 && pair_here[j - 1];
        const int a = second ? j - 1 : j;               // column with the real parts
        const bool two = second || (j < 2 && pair_here[j]);
        const int b = second ? j : (j < 2 ? j + 1 : j);  // column with the imaginary parts
        float x[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float rp = a == 0 ? ev[r][0] : (a == 1 ? ev[r][1] : ev[r][2]);
            const float ip = two ? (b == 1 ? ev[r][1] : ev[r][2]) : 0.0f;
            x[r] = rp * rp + ip * ip;
        }
        const float z = x[0] + (x[1] + x[2]);
        const float d = sqrtf(z);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float rp = a == 0 ? ev[r][0] : (a == 1 ? ev[r][1] : ev[r][2]);
            const float ip0 = two ? (b == 1 ? ev[r][1] : ev[r][2]) : 0.0f;
            const float ip = second ? -ip0 : ip0;  // the second column of a pair is the conjugate
            // normalize() divides the complex column by (d, 0): rows 0 and 1 as ONE Packet2cf -- SSE's pdiv = pmul(a, pconj(b)) / |b|^2 --, row 2
            // as a scalar complex division (libgcc's __divsc3, Smith); see oracle/eigensolver3f.h: normalized_real_part
            float q;
            if (r < 2) {
                q = (rp * d + (-(ip * -0.0f))) / (d * d + 0.0f * 0.0f);
            } else {
                const float ratio = 0.0f / d, denom = 0.0f * ratio + d;
                q = (ip * ratio + rp) / denom;
            }
            v[r][j] = z > 0.0f ? q : rp;
        }
    }
    return 0;
}
// Gaussians::limitCovariance (Gaussians.h:181-201)
__device__ void limit_covariance_f(float c[3][3]) {
    float v[3][3], lam[3];
    if (eigensolver3f(c, v, lam, nullptr) != 0) {  // not a state the reference defines (its solver's members stay uninitialised): keep the diagonal
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lam[i] = c[i][i];
#pragma unroll
            for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0f : 0.0f;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) lam[k] = lam[k] < 0.0001f ? 0.0001f : lam[k];  // std::max(eigenValues(k), 0.0001f), :191-194
    // eigenVectors * diagonal_matrix * eigenVectors.inverse() (Gaussians.h:200): the cofactor inverse of V, not its transpose
    float vinv[3][3];
    inverse3_f(v, vinv);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i][j] = sum3f((v[i][0] * lam[0]) * vinv[0][j], (v[i][1] * lam[1]) * vinv[1][j], (v[i][2] * lam[2]) * vinv[2][j]);
}
__device__ void inverse3_f(const float m[3][3], float inv[3][3]) {
    float cof[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            cof[i][j] = m[i1][j1] * m[i2][j2] - m[i1][j2] * m[i2][j1];
        }
    const float det = sum3f(cof[0][0] * m[0][0], cof[1][0] * m[1][0], cof[2][0] * m[2][0]);
    const float invdet = 1.0f / det;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) inv[r][c] = cof[c][r] * invdet;
}

// default path: `s` = centered^T * centered in Eigen's float order (fit_tree_group), divided by float(n - 1) in float (Gaussians.h:147)
__device__ void finish_gaussian_f(const float* __restrict__ s, int n, float* o) {
    const float fd = (float)(n - 1);
    float cov[3][3], inv[3][3];
    cov[0][0] = s[0] / fd, cov[0][1] = cov[1][0] = s[1] / fd, cov[0][2] = cov[2][0] = s[2] / fd;
    cov[1][1] = s[3] / fd, cov[1][2] = cov[2][1] = s[4] / fd, cov[2][2] = s[5] / fd;
    limit_covariance_f(cov);
    inverse3_f(cov, inv);
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) o[3 * c + r] = inv[r][c];
    o[9] = 0.0f, o[10] = (float)n, o[11] = 0.0f;
}

// Default path: the fit's sums in the order the oracle states (Gaussians::addPointSet there).
//  * subset.colwise().mean() (Gaussians.h:146) in Eigen's OWN order: the linear vectorised redux of a contiguous float column is a
//    pure function of the member count and of the column's offset in its 16-byte aligned buffer (oracle: eigen_linear_sum_f32; SSE2
//    Packet4f).  Per column: eight float chains (two packet accumulators of four lanes) over the aligned middle, res0 += res1, one odd
//    packet, predux (a0 + a2) + (a1 + a3), then the scalars in front of the aligned start and behind the last packet.  The chains
//    are serial, so the members' coordinates go through LDS in chunks: every wave of the group transforms members, 24 lanes (8 chains
//    x 3 columns) add them up, 3 lanes finish.
//  * centered^T * centered (Gaussians.h:147) in Eigen's OWN order as well (oracle: Gaussians::gemm_dot_f32, which cites the evaluators): below
//    14 members the lazy product -- every coefficient the linear redux of the element-wise products from an aligned start --, else the
//    blocked product, where 3 rows and 3 columns leave everything to gebp's scalar tail: per coefficient a float chain C = C + a_k b_k
//    over one depth block of kc members, res = res + C block after block, divided by float(n - 1) in float.  kc follows from the L1
//    data cache of the machine the reference runs on (dmsa_debug_options::eigen_l1_bytes; 680 for 32 KB), and the chains of different
//    blocks are independent: lane = (block, coefficient), every lane runs ONE chain of at most kc adds over the centred coordinates in
//    LDS.  A Gaussian that fits the group's LDS chunk is centred in place; a longer one goes through LDS again in slices that hold the
//    same W members of every block, so that all its block chains advance together (the members of the next slice are fetched into
//    registers while the chains of the current one run).
// Members are read from the Gaussian-ordered copy of the LOCAL points and transformed with the base pose table (the same
// operation sequence as k_transform, so the coordinates are the ones the voxelisation saw).
// The classes of the size-ordered Gaussian list (serial_kernels.h): 0 = long (16 waves per Gaussian), 1 = middle (4), 2 = short (1).
// The kernels read the class ranges from DEVICE memory, so they can be launched before the host knows the counts (with the
// previous iteration's counts as the grid): a workgroup beyond the true count exits.
__device__ __forceinline__ bool fit_task(const int32_t* __restrict__ sc /* n_chain, n_small, max, n_long */, int cls, int task, int& index) {
    const int n_chain = sc[0], n_small = sc[1], n_long = sc[3];
    const int first = cls == 0 ? 0 : (cls == 1 ? n_long : n_chain);
    const int count = cls == 0 ? n_long : (cls == 1 ? n_chain - n_long : n_small);
    index = first + task;
    return task < count;
}
// Eigen's linear redux of n contiguous floats that start `offset` floats behind a 16-byte boundary (Redux.h, LinearVectorizedTraversal):
// [0, aS) scalars in front, [aS, aEnd2) pairs of packets (the eight chains), [aEnd2, aEnd) one odd packet, [aEnd, n) scalars behind
struct ReduxPlan {
    int aS, aEnd2, aEnd;
    bool vec;  // at least one whole packet; otherwise a plain scalar loop
};
__device__ __forceinline__ ReduxPlan redux_plan(int n, int offset) {
    ReduxPlan p;
    p.aS = min((4 - (offset & 3)) & 3, n);
    const int rem = n - p.aS;
    p.vec = rem >= 4;
    p.aEnd = p.aS + (rem & ~3), p.aEnd2 = p.aS + (rem & ~7);
    return p;
}
// where element i of such a vector goes: -1 = chain (i - aS) & 7; otherwise a slot of the (at most ten) scalars the finish adds itself
__device__ __forceinline__ int redux_slot(const ReduxPlan& p, int i) {
    if (!p.vec) return i;  // n <= 6
    if (i < p.aS) return i;
    if (i >= p.aEnd2) return 3 + (i - p.aEnd2);
    return -1;
}
__device__ __forceinline__ ReduxPlan my_plan_col(int n, int c) { return redux_plan(n, c * (n & 3)); }  // column c of an n x 3 column-major matrix
constexpr int kReduxSlots = 10;
// chain `k` (0..7) of the plan over the elements [m0, m0 + cnt) held in sx[0 .. cnt): acc += every eighth element, in order.  The adds are
// a dependent chain (~8 cycles each); the LDS reads of the next eight elements are issued before the adds of the current eight, so the
// chain never waits for LDS.
__device__ __forceinline__ float redux_chain(const ReduxPlan& p, int k, int m0, int cnt, const float* __restrict__ sx, float acc) {
    const int first = p.aS + k;
    int i = m0 <= first ? first : first + ((m0 - first + 7) >> 3) * 8;
    const int end = min(m0 + cnt, p.aEnd2);
    if (i + 56 < end) {
        float v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sx[i - m0 + 8 * u];
        i += 64;
        for (; i + 56 < end; i += 64) {
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = sx[i - m0 + 8 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = acc + v[u];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = w[u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; i < end; i += 8) acc = acc + sx[i - m0];
    return acc;
}
// the tail of the redux: chains c and c + 4 meet, the odd packet, predux, scalars in front and behind (acc8: the eight chains, sp: the slots)
__device__ __forceinline__ float redux_finish(const ReduxPlan& p, int n, const float* __restrict__ acc8, const float* __restrict__ sp) {
    float res;
    if (!p.vec) {
        res = -0.0f;  // (-0) + x == x for every x: the first element is taken as it is
        for (int i = 0; i < n; ++i) res = res + sp[i];
        return res;
    }
    float r[4];
    const bool pairs = p.aEnd2 > p.aS;
#pragma unroll
    for (int l = 0; l < 4; ++l) r[l] = pairs ? acc8[l] + acc8[4 + l] : sp[3 + l];
    if (pairs && p.aEnd > p.aEnd2) {
#pragma unroll
        for (int l = 0; l < 4; ++l) r[l] = r[l] + sp[3 + l];
    }
    res = (r[0] + r[2]) + (r[1] + r[3]);
    for (int i = 0; i < p.aS; ++i) res = res + sp[i];
    for (int i = p.aEnd; i < n; ++i) res = res + sp[3 + (i - p.aEnd2)];
    return res;
}
// depth blocking of Eigen's (3 x n) * (n x 3) product (evaluateProductBlockingSizesHeuristic, one thread; oracle: Gaussians::gemm_kc)
__device__ __forceinline__ int gemm_kc(int k, int max_kc) {
    if (k < 48 || k <= max_kc) return k;
    const int r = k % max_kc;
    return r == 0 ? max_kc : max_kc - 8 * ((max_kc - 1 - r) / (8 * (k / max_kc + 1)));
}
// one coefficient of one depth block: C = C + a[k] * b[k], k = 0 .. cnt - 1, operands in LDS.  The adds are a dependent chain; the reads
// and products of the next eight steps are issued before the adds of the current eight.
// The products a_k * b_k are made by ALL threads of the group (rounded to float one by one, as the scalar loop does) and laid out per
// coefficient in LDS; a chain lane then only adds.  sp is 16-byte aligned: the operands of eight steps arrive as two ds_read_b128, in
// flight while the eight dependent adds of the previous eight run (sched_barrier pins that order: left alone, the compiler waits for
// the reads where it issues them).
__device__ __forceinline__ float cov_chain(const float* __restrict__ sp, int cnt, float C) {
    const float4* __restrict__ p4 = reinterpret_cast<const float4*>(sp);
    auto add8 = [&](const float4& a, const float4& b) { C = C + a.x, C = C + a.y, C = C + a.z, C = C + a.w, C = C + b.x, C = C + b.y, C = C + b.z, C = C + b.w; };
    int i = 0;
    if (cnt >= 8) {
        float4 x0 = p4[0], x1 = p4[1], y0, y1;  // two register sets take turns (no copies on the chain's path)
        i = 8;
        while (true) {
            const bool more_y = i + 8 <= cnt;
            if (more_y) y0 = p4[i >> 2], y1 = p4[(i >> 2) + 1], i += 8;
            __builtin_amdgcn_sched_barrier(0);
            add8(x0, x1);
            __builtin_amdgcn_sched_barrier(0);
            if (!more_y) break;
            const bool more_x = i + 8 <= cnt;
            if (more_x) x0 = p4[i >> 2], x1 = p4[(i >> 2) + 1], i += 8;
            __builtin_amdgcn_sched_barrier(0);
            add8(y0, y1);
            __builtin_amdgcn_sched_barrier(0);
            if (!more_x) break;
        }
    }
    for (; i < cnt; ++i) C = C + sp[i];
    return C;
}
// the coefficient-based lazy product of fewer than 14 members: the linear vectorised redux of the products from element 0 (Redux.h)
__device__ __forceinline__ float cov_lazy(const float* __restrict__ sp, int n) {
    const int packets = n >> 2;
    float res;
    if (packets == 0) {
        res = sp[0];
        for (int i = 1; i < n; ++i) res = res + sp[i];
        return res;
    }
    float r[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) r[l] = sp[l];
    if (packets > 1) {  // 8 <= n < 14: one pair of packets, then one odd packet from 12 members on
#pragma unroll
        for (int l = 0; l < 4; ++l) r[l] = r[l] + sp[4 + l];
        if (packets > 2) {
#pragma unroll
            for (int l = 0; l < 4; ++l) r[l] = r[l] + sp[8 + l];
        }
    }
    res = (r[0] + r[2]) + (r[1] + r[3]);
    for (int i = 4 * packets; i < n; ++i) res = res + sp[i];
    return res;
}
// covariance -> limitCovariance -> inverse (Gaussians.h:146-168, :181-201), one thread per Gaussian: the serial 3x3 work of all
// Gaussians fills whole waves instead of trailing every fit workgroup on a single lane
__global__ __launch_bounds__(256) void k_gauss_fit_finish(const int32_t* __restrict__ seg_off, const GaussCounts* __restrict__ counts,
                                                          const float* __restrict__ sums, float* __restrict__ info12) {
    const int M = counts->level[0].num_gauss + counts->level[1].num_gauss;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M) return;
    float a[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) a[q] = sums[(size_t)g * 6 + q];
    float o[12];
    finish_gaussian_f(a, seg_off[g + 1] - seg_off[g], o);
    float* dst = info12 + (size_t)g * 12;
#pragma unroll
    for (int i = 0; i < 12; ++i)
        if (i != 9) dst[i] = o[i];  // slot 9 is the rebalancing weight (k_size_classes / k_rebalancing_weights write it)
}
// All three classes -- and the rebalancing weights -- in ONE launch of 1024-thread workgroups: a workgroup is one long Gaussian (16 waves),
// four middle ones (4 waves each) or sixteen short ones (1 wave each); the first workgroup computes the weights.  Nothing has to be
// forked to a second stream and joined again (a cross-stream dependency costs ~15 us each way on this GPU,
// scripts/microbench/graph_edge.hip), and the weights do not sit behind the long fit.
// scratch of one group (a Gaussian's 16 / 4 / 1 waves) besides the block sums
struct FitScratch {
    float acc[3][8];              // the eight chains of each column
    float sp[3][kReduxSlots];     // scalars in front of / behind the packets, and the odd packet
    float mean[3];
    int rows[2];                  // smallest / largest pose-table row of the members (identity row excluded)
};
// block chains a group runs side by side: one lane per (block, coefficient), and at least 16 members of every block in the LDS chunk
#ifdef DMSA_FIT_TIMING
__device__ long long* g_fit_tim = nullptr;  // experiment build: per-workgroup phase times in pinned host memory (launch_gauss_fit_all prints them)
#endif
template <int kFitWaves>
struct FitGroup {
    static constexpr int kThreads = 64 * kFitWaves, CH = 256 * kFitWaves, CHS = CH + 8, kMaxG = (kThreads / 6) < (CH / 16) ? (kThreads / 6) : (CH / 16);
};
template <int kFitWaves>
__device__ __forceinline__ void fit_tree_group(const float4* __restrict__ memb_local, const int32_t* __restrict__ seg_off, const float4* __restrict__ table0,
                                               int g /* -1: no Gaussian for this group */, int gtid, float (*s_C)[6] /* [FitGroup::kMaxG] */, FitScratch* fs,
                                               float* sx /* [6][CHS]: pass 1 uses three columns */, int* s_rounds /* [3]: chunks, block groups, slices */, int id_row, int max_kc,
                                               float* __restrict__ memb_q /* [3][q_stride]: global coordinates of the members of Gaussians too long for LDS */,
                                               size_t q_stride, float* __restrict__ sums, int2* __restrict__ gauss_rows) {
    constexpr int CH = FitGroup<kFitWaves>::CH;    // members per chunk of the mean pass: four per lane
    constexpr int CHS = FitGroup<kFitWaves>::CHS;  // column stride in LDS: the three columns of one member lie in different banks
    constexpr int kThreads = FitGroup<kFitWaves>::kThreads, kMaxG = FitGroup<kFitWaves>::kMaxG;
    const int b = g >= 0 ? seg_off[g] : 0, n = g >= 0 ? seg_off[g + 1] - b : 0;
    const int lane = gtid & 63;
    // A group of ONE wave (the short class: sixteen Gaussians per workgroup) needs no workgroup barrier at all -- its LDS traffic is in
    // order, its scratch is its own -- so the sixteen waves run and end independently instead of meeting ten times at the pace of the
    // slowest.  Groups of several waves meet at the same barriers: everybody runs as many rounds as the group with the most.
    constexpr bool kSolo = kFitWaves == 1;
    // The barriers between phases order LDS traffic only: s_waitcnt lgkmcnt(0) + s_barrier.  __syncthreads() would also wait for every
    // global access in flight -- the members fetched for the next chunk while this one's chains run, the coordinates stored for pass 2 --,
    // i.e. one memory round trip per phase.  (What pass 2 reads back from global memory is fenced once, in front of pass 2.)
    auto sync = [&]() {
        if (kSolo) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    // the product's blocking (see above): kc members per depth block, nb blocks; Gg block chains per round, in slices of W members
    const bool lazy = n < 14;
    const bool resident = n <= CH;  // the coordinates of every member are still in LDS after the mean pass
    const int kc = lazy ? max(n, 1) : gemm_kc(n, max_kc), nb = lazy ? 0 : (n + kc - 1) / kc;
    const int Gg = max(min(nb, kMaxG), 1);
    // slices of a Gaussian that does not fit: the same W members of every block of a round, block t at t * (W + 4) -- W the largest
    // power of two with Gg * (W + 4) <= CH (>= 8); the stride keeps the 16-byte alignment of the chains' reads and spreads the blocks
    // over the banks
    const int logW = resident ? 0 : 31 - __clz(CH / Gg - 4), W = resident ? kc : 1 << logW, WS = W + 4;
    const int my_groups = lazy ? 0 : (nb + Gg - 1) / Gg, my_slices = resident ? 1 : (kc + W - 1) / W;
    if (!kSolo && gtid == 0) atomicMax(&s_rounds[0], (n + CH - 1) / CH), atomicMax(&s_rounds[1], my_groups), atomicMax(&s_rounds[2], my_slices);
    if (gtid == 0) fs->rows[0] = INT_MAX, fs->rows[1] = -1;
    sync();
    const int chunks = kSolo ? (n + CH - 1) / CH : s_rounds[0];
    const int groups = kSolo ? my_groups : s_rounds[1], slices = kSolo ? my_slices : s_rounds[2];
#ifdef DMSA_FIT_TIMING
    const long long t_start = wall_clock64();
    long long t_tr = 0, t_ch = 0, t_mark = t_start;
#define FIT_MARK(acc) { const long long t_now = wall_clock64(); acc += t_now - t_mark; t_mark = t_now; }
#else
#define FIT_MARK(acc)
#endif
    // ---- pass 1: subset.colwise().mean() in Eigen's order (see above) ----
    ReduxPlan plan[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) plan[c] = redux_plan(n, c * (n & 3));
    const int cc = gtid >> 3, ck = gtid & 7;  // chain lanes: gtid < 24 = (column, chain)
    const ReduxPlan my_plan = redux_plan(n, min(cc, 2) * (n & 3));  // (no run-time index into plan[]: that would put it into scratch memory)
    float acc = -0.0f;                         // (-0) + x == x: the chain starts with its first element as it is
    int rmin = INT_MAX, rmax = -1;
    float4 p[4];
    auto fetch_chunk = [&](int ch) {
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = memb_local[b + min(ch * CH + u * kThreads + gtid, max(n - 1, 0))];
    };
    if (chunks > 0) fetch_chunk(0);
    for (int ch = 0; ch < chunks; ++ch) {
        const int m0 = ch * CH;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int il = u * kThreads + gtid, i = m0 + il;
            if (i < n) {
                const int row = __float_as_int(p[u].w);
                if (row != id_row) rmin = min(rmin, row), rmax = max(rmax, row);
                const float3 q = apply_row3(table0[3 * row], table0[3 * row + 1], table0[3 * row + 2], p[u].x, p[u].y, p[u].z);
                sx[il] = q.x, sx[CHS + il] = q.y, sx[2 * CHS + il] = q.z;
                if (!resident) memb_q[b + i] = q.x, memb_q[q_stride + b + i] = q.y, memb_q[2 * q_stride + b + i] = q.z;  // pass 2 reads them back in slices
                if (i < 3 || i + 10 >= n || n <= 6) {  // one of the few scalars of some column's redux
                    int slot = redux_slot(plan[0], i);
                    if (slot >= 0) fs->sp[0][slot] = q.x;
                    slot = redux_slot(plan[1], i);
                    if (slot >= 0) fs->sp[1][slot] = q.y;
                    slot = redux_slot(plan[2], i);
                    if (slot >= 0) fs->sp[2][slot] = q.z;
                }
            }
        }
        sync();
        FIT_MARK(t_tr)
        if (ch + 1 < chunks) fetch_chunk(ch + 1);  // in flight while the chains of this chunk run
        if (gtid < 24) acc = redux_chain(my_plan, ck, m0, min(CH, n - m0), sx + cc * CHS, acc);
        if (ch + 1 < chunks) sync();  // the next chunk overwrites the coordinates
        FIT_MARK(t_ch)
    }
#ifdef DMSA_FIT_TIMING
    const long long t_p1 = wall_clock64();
    long long t_tr2 = 0, t_ch2 = 0;
#endif
    if (gtid < 24) fs->acc[cc][ck] = acc;
    {   // pose-table rows of the members (what evaluations of a Jacobian batch can differ from evaluation 0 for this Gaussian)
        rmin = wave_allmin(rmin), rmax = -wave_allmin(-rmax);
        if (lane == 0 && rmax >= 0) {
            if (kSolo)
                fs->rows[0] = rmin, fs->rows[1] = rmax;
            else
                atomicMin(&fs->rows[0], rmin), atomicMax(&fs->rows[1], rmax);
        }
    }
    sync();
    if (gtid < 3 && n > 0) fs->mean[gtid] = redux_finish(my_plan_col(n, gtid), n, fs->acc[gtid], fs->sp[gtid]) / (float)n;
    if (gtid == 3 && g >= 0 && gauss_rows != nullptr) gauss_rows[g] = make_int2(fs->rows[0], fs->rows[1]);
    sync();
    // ---- pass 2: centered^T * centered, xx xy xz yy yz zz, in the order of Eigen's product kernels (see above) ----
    const float mx = fs->mean[0], my = fs->mean[1], mz = fs->mean[2];
    const int cl_b = gtid / 6, cl_q = gtid - 6 * cl_b;  // chain lane = (block of the round, coefficient)
    // the six products of one member (MatrixXf centered = subset.rowwise() - mean, Gaussians.h:146; then a_k * b_k of the scalar loop) at position `at`
    auto put_products = [&](int at, float x, float y, float z) {
        const float cx = x - mx, cy = y - my, cz = z - mz;
        sx[at] = cx * cx, sx[CHS + at] = cx * cy, sx[2 * CHS + at] = cx * cz, sx[3 * CHS + at] = cy * cy, sx[4 * CHS + at] = cy * cz, sx[5 * CHS + at] = cz * cz;
    };
    // members of slice `sl` of round `bg`: position idx = (block idx >> logW of the round, member sl * W + (idx & (W - 1)) of the block)
    auto slot_member = [&](int bg, int sl, int idx) {
        const int bl = idx >> logW, k = sl * W + (idx & (W - 1)), blk = bg * Gg + bl, j = blk * kc + k;
        return (!resident && bl < Gg && blk < nb && k < kc && j < n) ? j : -1;
    };
    auto fetch_slice = [&](int bg, int sl) {  // (p[] is free after pass 1: x, y, z of the slot's member)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = slot_member(bg, sl, u * kThreads + gtid);
            if (j >= 0) p[u].x = memb_q[b + j], p[u].y = memb_q[q_stride + b + j], p[u].z = memb_q[2 * q_stride + b + j];
        }
    };
    if (!kSolo) {  // the coordinates other threads of the group stored in pass 1 must have landed before pass 2 reads them back
        __threadfence_block();
        __syncthreads();
    }
    if (groups > 0) fetch_slice(0, 0);
    if (resident) {  // the coordinates are still in LDS: every thread turns its own members into products, in place
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int il = u * kThreads + gtid;
            if (il < n) put_products(il, sx[il], sx[CHS + il], sx[2 * CHS + il]);
        }
    }
    sync();
    float res = 0.0f;  // dst.setZero() in front of the blocked product
    if (lazy && gtid < 6 && n > 0) res = cov_lazy(sx + gtid * CHS, n);
    for (int bg = 0; bg < groups; ++bg) {
        const int blk = bg * Gg + cl_b;
        const bool chainer = bg < my_groups && cl_b < Gg && blk < nb;
        const int len = chainer ? min(kc, n - blk * kc) : 0;
        float C = 0.0f;  // ResScalar C0(0)
        for (int sl = 0; sl < slices; ++sl) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = u * kThreads + gtid;
                if (slot_member(bg, sl, idx) >= 0) put_products((idx >> logW) * WS + (idx & (W - 1)), p[u].x, p[u].y, p[u].z);
            }
            sync();
            FIT_MARK(t_tr2)
            // the next slice is on its way while the chains of this one run
            if (sl + 1 < slices)
                fetch_slice(bg, sl + 1);
            else if (bg + 1 < groups)
                fetch_slice(bg + 1, 0);
            if (chainer && sl < my_slices) C = cov_chain(sx + cl_q * CHS + (resident ? blk * kc : cl_b * WS), min(max(len - sl * W, 0), W), C);
            sync();  // the next slice overwrites the products
            FIT_MARK(t_ch2)
        }
        if (chainer) s_C[cl_b][cl_q] = C;
        sync();
        if (gtid < 6 && bg < my_groups)  // res(i, j) += alpha * C0, block after block (alpha = 1)
            for (int t = 0; t < Gg && bg * Gg + t < nb; ++t) res = res + s_C[t][gtid];
        sync();
    }
    if (g >= 0 && gtid < 6) sums[(size_t)g * 6 + gtid] = res;
#ifdef DMSA_FIT_TIMING
    if (gtid == 0 && g_fit_tim != nullptr && (kFitWaves != 1 || (blockIdx.x & 31) == 0) && blockIdx.x < 2000) {  // 100 MHz clock: 10 ns per tick
        long long* r = g_fit_tim + 12 * (size_t)blockIdx.x;
        r[0] = kFitWaves, r[1] = n, r[2] = chunks, r[3] = groups, r[4] = slices, r[5] = wall_clock64() - t_start, r[6] = t_p1 - t_start, r[7] = t_tr, r[8] = t_ch,
        r[9] = wall_clock64() - t_p1, r[10] = t_tr2, r[11] = t_ch2;
    }
#endif
}
__device__ void rebalancing_weights_mirror_body(const int32_t* __restrict__ seg_off, GaussCounts* __restrict__ counts, float* __restrict__ info12, float* sx /* [8192] */,
                                                FitScratch* fs, const uint32_t* __restrict__ pow_codes, int pow_n);
struct FitLaunch {
    int first[3], tasks[3];  // per class: index of the first Gaussian of the class covered by this launch, number covered
    int wg[3];               // workgroups per class
    int weights;             // 1: workgroup 0 computes the rebalancing weights
    int id_row;              // the pose-table row of the static points (identity)
    int max_kc;              // largest depth block of Eigen's product on the reference's machine (from its L1 size)
    int pow_n;               // pow_codes covers the member counts 0 .. pow_n - 1
    const uint32_t* pow_codes;
};
constexpr int kFitLdsFloats = 16 * 6 * FitGroup<1>::CHS;  // dynamic LDS of k_gauss_fit_all: [group][x y z | six products][256 x waves + 8] -- 99 KB: one workgroup per CU, which its registers allow anyway
#ifndef DMSA_FIT_WAVES_PER_SIMD
#define DMSA_FIT_WAVES_PER_SIMD 4  // 8: two workgroups per CU (64 VGPRs, spills); 4: one (128 VGPRs) -- measured: scripts/ab_fit.sh
#endif
__global__ __launch_bounds__(1024, DMSA_FIT_WAVES_PER_SIMD) void k_gauss_fit_all(const float4* __restrict__ memb_local, const int32_t* __restrict__ seg_off, const float4* __restrict__ table0,
                                                        const uint32_t* __restrict__ order, const int32_t* __restrict__ sc, FitLaunch fl, float* __restrict__ sums,
                                                        GaussCounts* __restrict__ counts, float* __restrict__ info12, int2* __restrict__ gauss_rows,
                                                        float* __restrict__ memb_q, size_t q_stride) {
    extern __shared__ float4 s_x4[];  // kFitLdsFloats floats
    float* s_x = reinterpret_cast<float*>(s_x4);
    __shared__ float s_C[176][6];     // block sums of a round: class 0: [170]; class 1: 4 x [42]; class 2: 16 x [10]
    __shared__ FitScratch s_fs[16];
    __shared__ int s_rounds[3];
    if (threadIdx.x < 3) s_rounds[threadIdx.x] = 0;
    __syncthreads();
    const int tid = threadIdx.x;
    int bx = blockIdx.x;
    // the weights first: their chains (M / 8 dependent float adds) then run beside the fit of the longest Gaussians instead of behind it
    if (fl.weights) {
        if (bx == 0) {
            rebalancing_weights_mirror_body(seg_off, counts, info12, s_x, &s_fs[0], fl.pow_codes, fl.pow_n);
            return;
        }
        bx -= 1;
    }
    auto pick = [&](int cls, int task) {
        int index;
        return task < fl.tasks[cls] && fit_task(sc, cls, fl.first[cls] + task, index) ? (int)order[index] : -1;
    };
    if (bx < fl.wg[0]) {
        if (pick(0, bx) < 0) return;  // a surplus workgroup of the speculative launch
        fit_tree_group<16>(memb_local, seg_off, table0, pick(0, bx), tid, s_C, &s_fs[0], s_x, s_rounds, fl.id_row, fl.max_kc, memb_q, q_stride, sums, gauss_rows);
        return;
    }
    bx -= fl.wg[0];
    if (bx < fl.wg[1]) {
        const int grp = tid >> 8;
        fit_tree_group<4>(memb_local, seg_off, table0, pick(1, bx * 4 + grp), tid & 255, s_C + grp * FitGroup<4>::kMaxG, &s_fs[grp], s_x + grp * (6 * FitGroup<4>::CHS), s_rounds, fl.id_row,
                          fl.max_kc, memb_q, q_stride, sums, gauss_rows);
        return;
    }
    bx -= fl.wg[1];
    if (bx < fl.wg[2]) {
        const int grp = tid >> 6;
        fit_tree_group<1>(memb_local, seg_off, table0, pick(2, bx * 16 + grp), tid & 63, s_C + grp * FitGroup<1>::kMaxG, &s_fs[grp], s_x + grp * (6 * FitGroup<1>::CHS), s_rounds, fl.id_row,
                          fl.max_kc, memb_q, q_stride, sums, gauss_rows);
        return;
    }
}
static_assert(FitGroup<16>::kMaxG <= 176 && 4 * FitGroup<4>::kMaxG <= 176 && 16 * FitGroup<1>::kMaxG <= 176, "s_C of k_gauss_fit_all");
static_assert(6 * FitGroup<16>::CHS <= kFitLdsFloats && 4 * 6 * FitGroup<4>::CHS <= kFitLdsFloats, "dynamic LDS of k_gauss_fit_all");
void launch_gauss_fit_all(const float4* memb_local, const int32_t* seg_off, const float* table0, const uint32_t* order, const int32_t* sc, const int first[3],
                          const int tasks[3], float* sums, GaussCounts* counts, float* info12, bool with_weights, int id_row, int2* gauss_rows, int eigen_l1_bytes,
                          const uint32_t* pow_codes, int pow_n, float* memb_q, size_t q_stride, hipStream_t s) {
    FitLaunch fl;
    for (int c = 0; c < 3; ++c) fl.first[c] = first[c], fl.tasks[c] = tasks[c] > 0 ? tasks[c] : 0;
    fl.wg[0] = fl.tasks[0], fl.wg[1] = (fl.tasks[1] + 3) / 4, fl.wg[2] = (fl.tasks[2] + 15) / 16;
    fl.weights = with_weights ? 1 : 0, fl.id_row = id_row;
    // max_kc of evaluateProductBlockingSizesHeuristic: ((l1 - mr * nr * 4) / (mr * 4 + nr * 4)) & ~(k_peeling - 1) with mr = 8, nr = 4
    fl.max_kc = std::max(((eigen_l1_bytes - 128) / 48) & ~7, 1);
    fl.pow_codes = pow_codes, fl.pow_n = pow_codes ? pow_n : 0;
    const int grid = fl.wg[0] + fl.wg[1] + fl.wg[2] + fl.weights;
    if (grid <= 0) return;
#ifdef DMSA_FIT_TIMING
    {
        static long long* h_tim = nullptr;
        static int launches = 0;
        if (!h_tim) {
            (void)hipHostMalloc(reinterpret_cast<void**>(&h_tim), 2000 * 12 * 8, hipHostMallocDefault);
            std::memset(h_tim, 0, 2000 * 12 * 8);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fit_tim), &h_tim, sizeof(h_tim));
        } else if (++launches == 6) {  // print the records of one warm launch
            (void)hipStreamSynchronize(s);
            for (int w = 0; w < 2000; ++w) {
                const long long* r = h_tim + 12 * w;
                if (r[0]) std::fprintf(stderr, "fit wg %d waves %lld n %lld chunks %lld groups %lld slices %lld | total %lld pass1 %lld (produce %lld chain %lld) pass2 %lld (produce %lld chain %lld) x10ns\n",
                                       w, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]);
            }
        }
    }
#endif
    // 99 KB of dynamic LDS on top of ~8 KB static: above the 64 KB a kernel gets without asking (per launch: the attribute belongs to the device)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gauss_fit_all), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFitLdsFloats * sizeof(float)));
    hipLaunchKernelGGL(k_gauss_fit_all, dim3(grid), dim3(1024), kFitLdsFloats * sizeof(float), s, memb_local, seg_off, reinterpret_cast<const float4*>(table0), order, sc, fl,
                       sums, counts, info12, gauss_rows, memb_q, q_stride);
}
void launch_gauss_fit_finish(const int32_t* seg_off, const GaussCounts* counts, const float* sums, int max_gauss, float* info12, hipStream_t s) {
    if (max_gauss > 0) hipLaunchKernelGGL(k_gauss_fit_finish, dim3((max_gauss + 255) / 256), dim3(256), 0, s, seg_off, counts, sums, info12);
}

// numPointsPerSet.cast<float>().array().pow(-1) (Gaussians.h:172): Eigen promotes the int exponent and calls std::pow(float, float) per
// coefficient -- libm's powf, which is NOT correctly rounded (glibc >= 2.27: one ulp away from 1.0f / n for 953, 2071, 5331, ... --
// 0.06 % of the n < 2^24).  The host asks ITS libm (the one the reference would run on) once per count and hands the differences over
// as two bits per n: 0 same bits as the division, 1 one ulp above, 2 one ulp below (context.cpp: powm1_codes).  Counts beyond the
// table take the division.
__device__ __forceinline__ float pow_minus_one(int n, const uint32_t* __restrict__ pow_codes, int pow_n) {
    float w = 1.0f / (float)n;
    if (n < pow_n) {
        const uint32_t code = (pow_codes[n >> 4] >> (2 * (n & 15))) & 3u;
        w = __int_as_float(__float_as_int(w) + (code == 1u ? 1 : (code == 2u ? -1 : 0)));
    }
    return w;
}
__global__ void k_pow_minus_one(const int32_t* __restrict__ n, int count, const uint32_t* __restrict__ pow_codes, int pow_n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = pow_minus_one(n[i], pow_codes, pow_n);
}
// test hook (dmsa_debug_limit_covariance): Gaussians::limitCovariance and its EigenSolver on caller matrices (column-major 3 x 3 each)
__global__ __launch_bounds__(256) void k_debug_limit_covariance(const float* __restrict__ cov9, int64_t count, float* __restrict__ out9, float* __restrict__ evals3,
                                                                float* __restrict__ V9, int32_t* __restrict__ iters, int32_t* __restrict__ info) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= count) return;
    float c[3][3], v[3][3], lam[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) c[r][k] = cov9[9 * g + 3 * k + r];
    int it = 0;
    const int st = eigensolver3f(c, v, lam, &it);
    if (iters) iters[g] = it;
    if (info) info[g] = st;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (evals3) evals3[3 * g + k] = st == 0 ? lam[k] : 0.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (V9) V9[9 * g + 3 * k + r] = st == 0 ? v[r][k] : (r == k ? 1.0f : 0.0f);
    }
    limit_covariance_f(c);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) out9[9 * g + 3 * k + r] = c[r][k];
}
void launch_debug_limit_covariance(const float* cov9, int64_t count, float* out9, float* evals3, float* V9, int32_t* iters, int32_t* info, hipStream_t s) {
    if (count > 0) hipLaunchKernelGGL(k_debug_limit_covariance, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, cov9, count, out9, evals3, V9, iters, info);
}
void launch_pow_minus_one(const int32_t* n, int count, const uint32_t* pow_codes, int pow_n, float* out, hipStream_t s) {
    if (count > 0) hipLaunchKernelGGL(k_pow_minus_one, dim3((count + 255) / 256), dim3(256), 0, s, n, count, pow_codes, pow_n, out);
}
// default path: rebalancingWeights.head(M).mean() (Gaussians.h:176) in Eigen's own order -- the linear redux of M contiguous floats that
// start at their aligned buffer (see fit_tree_group): the weights go through LDS 8192 at a time, eight lanes carry the chains
__device__ void rebalancing_weights_mirror_body(const int32_t* __restrict__ seg_off, GaussCounts* __restrict__ counts, float* __restrict__ info12, float* sx /* [8192] */,
                                                FitScratch* fs, const uint32_t* __restrict__ pow_codes, int pow_n) {
    constexpr int CH = 8192;
    const int M = counts->level[0].num_gauss + counts->level[1].num_gauss;
    const ReduxPlan plan = redux_plan(M, 0);
    float acc = -0.0f;
    for (int m0 = 0; m0 < M; m0 += CH) {
#pragma unroll
        for (int u = 0; u < CH / 1024; ++u) {
            const int il = u * 1024 + threadIdx.x, g = m0 + il;
            if (g < M) {
                const float w = pow_minus_one(seg_off[g + 1] - seg_off[g], pow_codes, pow_n) * 1.0f;  // * obervationWeights (1.0, :175)
                sx[il] = w;
                info12[(size_t)g * 12 + 9] = w;
                const int slot = redux_slot(plan, g);
                if (slot >= 0) fs->sp[0][slot] = w;
            }
        }
        __syncthreads();
        if (threadIdx.x < 8) acc = redux_chain(plan, threadIdx.x, m0, min(CH, M - m0), sx, acc);
        __syncthreads();
    }
    if (threadIdx.x < 8) fs->acc[0][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        fs->mean[0] = redux_finish(plan, M, fs->acc[0], fs->sp[0]) / (float)M;
        counts->weight_mean = fs->mean[0];
    }
    __syncthreads();
    const float mean = fs->mean[0];
    for (int g = threadIdx.x; g < M; g += 1024) info12[(size_t)g * 12 + 9] = info12[(size_t)g * 12 + 9] / mean;
}
// ------------------------------------------------------------------------------------------------------------
// K5 — normal equations and squared-error sums (fp64, deterministic two-stage reductions)
// ------------------------------------------------------------------------------------------------------------
constexpr int kNeTile = 32;

static inline int ne_rows_per_split(int rows, int P) {
    int rs = 256;
    if (P > 64) {
        const int want = (rows + 31) / 32;  // at most 32 splits when the tile grid is already large
        rs = ((want + 31) / 32) * 32;
        if (rs < 256) rs = 256;
    }
    return rs;
}
int normal_equations_partial_doubles(int rows, int P) {
    const int nt = (P + 1 + kNeTile - 1) / kNeTile;
    const int rs = ne_rows_per_split(rows, P);
    const int nsplit = (rows + rs - 1) / rs;
    return nsplit * nt * nt * kNeTile * kNeTile;
}

// column k of A' = [J | e0] at row r
__device__ __forceinline__ double ne_col(const double* __restrict__ E, int64_t ldE, int P, double inv_h, int k, int r, int rows) {
    if (r >= rows || k > P) return 0.0;
    const double e0 = E[r];
    if (k == P) return e0;
    return inv_h * (E[(size_t)(k + 1) * ldE + r] - e0);
}

// A 32 x 32 output tile of a row block is shared by kNeQuarters workgroups (blockIdx.z = which 8 of the 32 second-operand columns): 256
// threads, one output each, so the row block's LDS traffic (the bound of this kernel: two 8-byte reads per multiply-add) is spread over
// four compute units.  The whole row block (rs = 256 rows = 8 stages for P <= 64) is fetched into registers up front: one memory
// latency instead of one per stage.  Every output sums its row block row by row.
constexpr int kNeQuarters = 4, kNeQCols = kNeTile / kNeQuarters;
__global__ __launch_bounds__(256) void k_normal_eq_partial(const double* __restrict__ E, int64_t ldE, int rows, int P, double inv_h, int rs, int nt,
                                                           double* __restrict__ partial) {
    __shared__ double s_a[kNeTile][kNeTile + 1];
    __shared__ double s_b[kNeQCols][kNeTile + 1];
    const int tile = blockIdx.x, ti = tile % nt, tj = tile / nt;
    const int split = blockIdx.y, quarter = blockIdx.z;
    const int tx = threadIdx.x & 31, tyl = threadIdx.x >> 5;  // output (column ti*32 + tx of A', column tj*32 + quarter*8 + tyl)
    const int rr = threadIdx.x & 31, kq = threadIdx.x >> 5;   // staging: row rr of the stage; columns kq, kq + 8, kq + 16, kq + 24 of the first operand
    double c = 0.0;
    const int r_begin = split * rs, r_end = min(rows, r_begin + rs);
    constexpr int kStages = 8;
    double na[kStages][kNeQuarters], nb[kStages];
    for (int g0 = r_begin; g0 < r_end; g0 += kStages * kNeTile) {
#pragma unroll
        for (int u = 0; u < kStages; ++u) {
            const int r = g0 + u * kNeTile + rr;
            const bool in = r < r_end;
#pragma unroll
            for (int q = 0; q < kNeQuarters; ++q) na[u][q] = in ? ne_col(E, ldE, P, inv_h, ti * kNeTile + kq + q * kNeQCols, r, rows) : 0.0;
            nb[u] = in ? ne_col(E, ldE, P, inv_h, tj * kNeTile + quarter * kNeQCols + kq, r, rows) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < kStages; ++u) {
            if (g0 + u * kNeTile >= r_end) break;
#pragma unroll
            for (int q = 0; q < kNeQuarters; ++q) s_a[kq + q * kNeQCols][rr] = na[u][q];
            s_b[kq][rr] = nb[u];
            __syncthreads();
#pragma unroll 8
            for (int q = 0; q < kNeTile; ++q) c += s_a[tx][q] * s_b[tyl][q];
            __syncthreads();
        }
    }
    partial[((size_t)split * nt * nt + tile) * kNeTile * kNeTile + (quarter * kNeQCols + tyl) * kNeTile + tx] = c;
}
__global__ __launch_bounds__(256) void k_normal_eq_reduce(const double* __restrict__ partial, int nsplit, int nt, int P, double* __restrict__ Hp) {
    const int n1 = P + 1;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n1 * n1) return;
    const int i = q % n1, j = q / n1;  // Hp col-major: element (i, j)
    const int ti = i / kNeTile, tj = j / kNeTile, li = i % kNeTile, lj = j % kNeTile;
    // block sums in block order (the oracle's rule); the loads of a batch are issued together, only the adds are a chain
    const double* src = partial + ((size_t)tj * nt + ti) * kNeTile * kNeTile + lj * kNeTile + li;
    const size_t stride = (size_t)nt * nt * kNeTile * kNeTile;
    double s = 0.0;
    int sp = 0;
    for (; sp + 16 <= nsplit; sp += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += v[u];
    }
    for (; sp < nsplit; ++sp) s += src[(size_t)sp * stride];
    Hp[(size_t)j * n1 + i] = s;
}
// P > 64 (keyframe pass: 2 * rows * P^2 is hundreds of MFLOP): the same 32 x 32 tiles on the matrix cores.  A workgroup = 2 x 2
// waves, each wave one 16 x 16 tile of D = A_i^T A_j through chained v_mfma_f64_16x16x4_f64: k = four consecutive ROWS, so the
// accumulator runs d = fma(a_r, b_r, d) row by row over the whole row block -- the order the oracle states for P > 64
// (its blocked_dot; the instruction's rounding was pinned with scripts/microbench/mfma_f64_semantics.hip:
// a fused chain, k ascending).  Operand panels (32 columns x 64 rows each) are staged through LDS with row-contiguous loads.
typedef double d4v __attribute__((ext_vector_type(4)));
constexpr int kMfmaRows = 64;
// columns of A' = [J | e0] in place: E[k + 1][r] <- inv_h * (E[k + 1][r] - E[0][r]) (DmsaOptimizer.h:226), E[0] stays e0.  The residual
// batch is consumed by the normal equations only, and every panel element is then ONE load for the 2 * nt tiles that need it.
// With `skip` (serial_kernels.h: launch_eval_row_ranges / launch_gauss_fit_all), the pairs (Gaussian r, evaluation k + 1) whose pose-table
// rows all have evaluation 0's bits were not computed: their residual IS E[0][r], so the column entry is inv_h * (e0 - e0).
constexpr int kJacColsPerThread = 8;   // evaluations per thread: e0 and the Gaussian's row range are loaded once for all of them
__global__ __launch_bounds__(256) void k_jacobian_columns(double* __restrict__ E, int64_t ldE, int rows, int P, double inv_h, const EvalSkip skip) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x, k0 = blockIdx.y * kJacColsPerThread;
    const bool in = r < rows, sparse = skip.row_range != nullptr && r < skip.M;
    const double e0 = in ? E[r] : 0.0;
    const int2 gr = sparse ? skip.gauss_rows[r] : make_int2(0, 0);
    double ek[kJacColsPerThread];
    bool same[kJacColsPerThread];
#pragma unroll
    for (int u = 0; u < kJacColsPerThread; ++u) {  // the loads of all eight evaluations in flight together
        const int k = k0 + u;
        same[u] = false;
        if (k < P && sparse) {
            const int2 e = skip.row_range[k + 1];
            same[u] = e.y < gr.x || e.x > gr.y;
        }
        ek[u] = e0;
        if (k < P && in && (!same[u] || skip.check)) ek[u] = E[(size_t)(k + 1) * ldE + r];
    }
#pragma unroll
    for (int u = 0; u < kJacColsPerThread; ++u) {
        const int k = k0 + u;
        if (k >= P) break;
        // eval_skip = 2: the pair WAS computed and must agree
        const bool bad = in && same[u] && skip.check && __double_as_longlong(ek[u]) != __double_as_longlong(e0);
        if (in) E[(size_t)(k + 1) * ldE + r] = inv_h * ((same[u] && !skip.check ? e0 : ek[u]) - e0);
        if (skip.stats != nullptr) {  // one counter pair per evaluation: thousands of waves adding to ONE address serialise (measured: +0.5 ms)
            const unsigned long long n_same = __popcll(__ballot(same[u])), n_bad = __popcll(__ballot(bad));
            if ((threadIdx.x & 63) == 0) {
                if (n_same) atomicAdd(&skip.stats[2 * k], n_same);
                if (n_bad) atomicAdd(&skip.stats[2 * k + 1], n_bad);
            }
        }
    }
}
// element (column c of A', row r): c < P -> E[c + 1][r], c == P -> E[0][r], beyond -> 0
__device__ __forceinline__ double ne_col_scaled(const double* __restrict__ E, int64_t ldE, int P, int c, int r) {
    return c > P ? 0.0 : E[(size_t)(c == P ? 0 : c + 1) * ldE + r];
}
// Only tiles with ti <= tj are computed (H is symmetric, and a fused chain of a_r * b_r does not care which factor is which); an
// off-diagonal workgroup writes its tile and the mirrored one.  The panels of the next 64 rows are loaded into registers before
// the MFMAs of the current ones.
__global__ __launch_bounds__(256) void k_normal_eq_mfma(const double* __restrict__ E, int64_t ldE, int rows, int P, int rs, int nt,
                                                        double* __restrict__ partial) {
    __shared__ double s_a[kNeTile][kMfmaRows + 4];
    __shared__ double s_b[kNeTile][kMfmaRows + 4];
    int ti = 0, tj = 0;  // blockIdx.x enumerates the pairs ti <= tj row by row
    for (int rest = blockIdx.x, row = 0; row < nt; ++row) {
        if (rest < nt - row) {
            ti = row, tj = row + rest;
            break;
        }
        rest -= nt - row;
    }
    const int split = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wi = wave & 1, wj = wave >> 1;  // the wave's 16 x 16 quarter of the 32 x 32 tile
    d4v acc = {0.0, 0.0, 0.0, 0.0};
    const int r_begin = split * rs, r_end = min(rows, r_begin + rs);
    constexpr int kPer = kNeTile * kMfmaRows / 256;  // panel elements per thread
    double ra[kPer], rb[kPer];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int q = threadIdx.x + u * 256, kk = q / kMfmaRows, r = r0 + q % kMfmaRows;
            const bool in = r < r_end;  // rows past the block: zeros, fma(0, 0, d) == d
            ra[u] = in ? ne_col_scaled(E, ldE, P, ti * kNeTile + kk, r) : 0.0;
            rb[u] = in ? ne_col_scaled(E, ldE, P, tj * kNeTile + kk, r) : 0.0;
        }
    };
    fetch(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += kMfmaRows) {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int q = threadIdx.x + u * 256;
            s_a[q / kMfmaRows][q % kMfmaRows] = ra[u], s_b[q / kMfmaRows][q % kMfmaRows] = rb[u];
        }
        __syncthreads();
        if (r0 + kMfmaRows < r_end) fetch(r0 + kMfmaRows);
        const double* pa = &s_a[wi * 16 + (lane & 15)][lane >> 4];
        const double* pb = &s_b[wj * 16 + (lane & 15)][lane >> 4];
#pragma unroll
        for (int k4 = 0; k4 < kMfmaRows / 4; ++k4)  // A[i][k] = column i at row r0 + 4 k4 + k, B[k][j] likewise
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[4 * k4], pb[4 * k4], acc, 0, 0, 0);
        __syncthreads();
    }
    double* out = partial + ((size_t)split * nt * nt + (size_t)tj * nt + ti) * kNeTile * kNeTile;
    double* mirror = partial + ((size_t)split * nt * nt + (size_t)ti * nt + tj) * kNeTile * kNeTile;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // register r of lane l: D[i = l / 16 + 4 r][j = l % 16]
        const int li = wi * 16 + (lane >> 4) + 4 * r, lj = wj * 16 + (lane & 15);
        out[lj * kNeTile + li] = acc[r];
        if (ti != tj) mirror[li * kNeTile + lj] = acc[r];
    }
}
NormalEqPartials normal_equations_partials(int rows, int P) {
    NormalEqPartials q;
    q.nt = (P + 1 + kNeTile - 1) / kNeTile;
    const int rs = ne_rows_per_split(rows, P);
    q.nsplit = (rows + rs - 1) / rs;
    return q;
}
void launch_normal_equations(const double* E, int64_t ldE, int rows, int P, double inv_h, double* partial, double* Hp, hipStream_t s, bool reduce, const EvalSkip* skip) {
    const int nt = (P + 1 + kNeTile - 1) / kNeTile;
    const int rs = ne_rows_per_split(rows, P);
    const int nsplit = (rows + rs - 1) / rs;
    if (P > 64) {  // NOTE: turns the residual batch E into the columns of [J | e0] in place
        hipLaunchKernelGGL(k_jacobian_columns, dim3((rows + 255) / 256, (P + kJacColsPerThread - 1) / kJacColsPerThread), dim3(256), 0, s, const_cast<double*>(E), ldE, rows, P, inv_h, skip ? *skip : EvalSkip{});
        hipLaunchKernelGGL(k_normal_eq_mfma, dim3(nt * (nt + 1) / 2, nsplit), dim3(256), 0, s, E, ldE, rows, P, rs, nt, partial);
    } else
        hipLaunchKernelGGL(k_normal_eq_partial, dim3(nt * nt, nsplit, kNeQuarters), dim3(256), 0, s, E, ldE, rows, P, inv_h, rs, nt, partial);
    const int n1 = P + 1;
    if (reduce) hipLaunchKernelGGL(k_normal_eq_reduce, dim3((n1 * n1 + 255) / 256), dim3(256), 0, s, partial, nsplit, nt, P, Hp);
}

// out[b] = the block sums of evaluation b added in block order
__global__ void k_squared_sums_reduce(const double* __restrict__ partial, int nsplit, int B, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) s += partial[(size_t)b * nsplit + sp];
    out[b] = s;
}
// Parity path: e^T e of every evaluation in the blocked order of the normal equations (blocks of ne_rows_per_split(rows, P)
// consecutive rows summed row by row, block sums added in order) -- the order the oracle states, so that the line search compares
// bit-identical numbers.
__global__ __launch_bounds__(64) void k_squared_sums_blocked(const double* __restrict__ E, int64_t ldE, int rows, int rs, int nsplit, int fused,
                                                             double* __restrict__ partial, const DevSync sy) {
    // one wave per (row block, evaluation): the block's values arrive by coalesced loads, lane 0 adds their squares row by row
    __shared__ double s_v[1024];
    const int sp = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (sy.wait_counter != nullptr) {  // the additional rows of the trial batch come from the side stream (optimize_loop.cpp); long there by now
        dev_sync_enter(sy);
        __syncthreads();
    }
    const int r0 = sp * rs, r_end = min(rows, r0 + rs);
    double s = 0.0;
    for (int c0 = r0; c0 < r_end; c0 += 1024) {
        const int cnt = min(1024, r_end - c0);
        for (int i = lane; i < cnt; i += 64) s_v[i] = E[(size_t)b * ldE + c0 + i];
        __syncthreads();
        if (lane == 0) {
            if (fused)  // P > 64: the rule of the matrix-core normal equations, one fused multiply-add per row
                for (int i = 0; i < cnt; ++i) s = fma(s_v[i], s_v[i], s);
            else
                for (int i = 0; i < cnt; ++i) {
                    const double v = s_v[i];
                    s += v * v;
                }
        }
        __syncthreads();
    }
    if (lane == 0) partial[(size_t)b * nsplit + sp] = s;
}
int squared_sums_blocked_partial_doubles(int rows, int P, int B) {
    const int rs = ne_rows_per_split(rows, P);
    return ((rows + rs - 1) / rs) * B;
}
void launch_squared_sums_blocked(const double* E, int64_t ldE, int rows, int P, int B, double* partial, double* out, hipStream_t s, const DevSync* sy) {
    const int rs = ne_rows_per_split(rows, P);
    const int nsplit = (rows + rs - 1) / rs;
    hipLaunchKernelGGL(k_squared_sums_blocked, dim3(nsplit, B), dim3(64), 0, s, E, ldE, rows, rs, nsplit, P > 64 ? 1 : 0, partial, sy ? *sy : DevSync{});
    if (out) hipLaunchKernelGGL(k_squared_sums_reduce, dim3((B + 63) / 64), dim3(64), 0, s, partial, nsplit, B, out);  // out null: the consumer adds the block sums
}

}  // namespace dmsa
