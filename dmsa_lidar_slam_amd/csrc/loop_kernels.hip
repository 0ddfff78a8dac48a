// loop_kernels.hip — see loop_kernels.h.  Small fp64 kernels (one wave per evaluation, lane = pose) that keep optimizeSet's control
// state on the device.  Everything here is latency-, not throughput-bound: the point is that the host no longer has to wait for the
// normal equations, solve, re-chain and upload between the two evaluation batches of an iteration.
//
// Operation order = host_math.cpp's (both sides compile pose_math.h): a chain is exp per pose (parallel over lanes), the running
// product R_k = R_{k-1} exp(o_k), T_k = T_{k-1} + R_{k-1} t_k strictly left to right on one lane (rounding makes it order dependent),
// then log per pose (parallel).
#include "loop_kernels.h"

#include <cfloat>

namespace dmsa {

namespace {

constexpr int kWave = 64;

__device__ __forceinline__ void store_mat(double* dst, const Mat3& M) {
#pragma unroll
    for (int i = 0; i < 9; ++i) dst[i] = M.a[i];
}
__device__ __forceinline__ Mat3 load_mat(const double* src) {
    Mat3 M;
#pragma unroll
    for (int i = 0; i < 9; ++i) M.a[i] = src[i];
    return M;
}

// relative2global on LDS arrays, one wave: ConsecutivePoses.h:26-43
__device__ void wave_relative_to_global(int n, const double* rel_o, const double* rel_t, double* glob_o, double* glob_t, double* sE, double* sR) {
    const int lane = threadIdx.x;
    for (int k = lane; k < n; k += kWave) store_mat(sE + 9 * k, so3_exp(col3(rel_o, k)));
    __syncthreads();
    if (lane == 0) {
        Mat3 R = Mat3::identity();
        Vec3 T{0, 0, 0};
        for (int k = 0; k < n; ++k) {
            T = T + R * col3(rel_t, k);
            set_col3(glob_t, k, T);
            R = R * load_mat(sE + 9 * k);
            store_mat(sR + 9 * k, R);
        }
    }
    __syncthreads();
    for (int k = lane; k < n; k += kWave) set_col3(glob_o, k, so3_log(load_mat(sR + 9 * k)));
    __syncthreads();
}
// global2relative: ConsecutivePoses.h:45-67 (every pose independent of the others)
__device__ void wave_global_to_relative(int n, const double* glob_o, const double* glob_t, double* rel_o, double* rel_t) {
    const int lane = threadIdx.x;
    for (int k = lane; k < n; k += kWave) {
        if (k == 0) {
            set_col3(rel_o, 0, col3(glob_o, 0));
            set_col3(rel_t, 0, col3(glob_t, 0));
        } else {
            Vec3 ro, rt;
            unchain_step(col3(glob_o, k - 1), col3(glob_t, k - 1), col3(glob_o, k), col3(glob_t, k), ro, rt);
            set_col3(rel_o, k, ro), set_col3(rel_t, k, rt);
        }
    }
    __syncthreads();
}

// Pose 0 of the window model under updateImuError's global2relative (ContinuousTrajectory.h:606): every evaluation replaces the relative
// pose 0 by the global pose 0 the chain just derived from it, p <- g(p) with g = (log(I exp(o)), 0 + I t).  Evaluation j of a batch
// starts from g^j(p0).  g reaches a fixed point (or a short cycle) after one or two applications, so the orbit is followed until it
// repeats instead of j times.
struct Pose0 {
    Vec3 o, t;
};
__device__ __forceinline__ Pose0 pose0_round_trip(const Pose0 p) {
    Mat3 R = Mat3::identity();
    Vec3 T{0, 0, 0};
    Vec3 go, gt;
    chain_step(R, T, p.o, p.t, go, gt);
    return Pose0{go, gt};
}
__device__ __forceinline__ bool same_bits(const Pose0& a, const Pose0& b) {
    return a.o.x == b.o.x && a.o.y == b.o.y && a.o.z == b.o.z && a.t.x == b.t.x && a.t.y == b.t.y && a.t.z == b.t.z;
}
__device__ Pose0 pose0_orbit(Pose0 p0, int j) {
    constexpr int kHist = 6;
    Pose0 hist[kHist];
    hist[0] = p0;
    Pose0 p = p0;
    for (int i = 1; i <= j; ++i) {
        p = pose0_round_trip(p);
        if (i < kHist) {
            for (int q = 0; q < i; ++q)
                if (same_bits(hist[q], p)) {  // the orbit is periodic from q on with period i - q
                    const int L = i - q;
                    return hist[q + (j - q) % L];
                }
            hist[i] = p;
        }
    }
    return p;
}

// additional rows of the CURRENT chain in LDS -> rows[0 .. extra); the window model's updateImuError runs global2relative first
__device__ void wave_extra_rows(const LoopModel& m, double* rel_o, double* rel_t, const double* glob_o, const double* glob_t, double* rows, double* dense) {
    const int lane = threadIdx.x;
    if (m.extra <= 0) return;
    if (m.model == 1) {
        wave_global_to_relative(m.n, glob_o, glob_t, rel_o, rel_t);
        // the twelve dense-trajectory values of every row on twelve lanes (independent Floater-Hormann evaluations, six divisions each),
        // then one lane per row: the same operations as imu_row()
        for (int i = lane; i < 12 * (m.n - 1); i += kWave) dense[i] = imu_dense_value(1 + i / 12, i % 12, m.n, m.stamps, m.fhw, m.traj_time, m.imu, glob_t);
        __syncthreads();
        for (int k = 1 + lane; k < m.n; k += kWave) rows[k - 1] = imu_row_from_dense(k, dense + 12 * (k - 1), m.stamps, m.imu, glob_o, glob_t, col3(rel_o, k));
    } else {
        int at = 0;
        if (m.key.use_gravity) {
            for (int k = lane; k < m.n; k += kWave) rows[k] = gravity_row(k, m.key, col3(glob_o, k));
            at = m.n;
        }
        if (m.key.use_odometry)
            for (int k = 1 + lane; k < m.n; k += kWave) rows[at + k - 1] = odometry_row(k, m.key, col3(rel_o, k), col3(rel_t, k));
    }
    __syncthreads();
}

__device__ __forceinline__ void load_state(const double* st, int n, double* rel_o, double* rel_t, double* glob_o, double* glob_t) {
    for (int i = threadIdx.x; i < 3 * n; i += kWave) rel_o[i] = st[i], rel_t[i] = st[3 * n + i], glob_o[i] = st[6 * n + i], glob_t[i] = st[9 * n + i];
    __syncthreads();
}
__device__ __forceinline__ void store_state(double* st, int n, const double* rel_o, const double* rel_t, const double* glob_o, const double* glob_t) {
    for (int i = threadIdx.x; i < 3 * n; i += kWave) st[i] = rel_o[i], st[3 * n + i] = rel_t[i], st[6 * n + i] = glob_o[i], st[9 * n + i] = glob_t[i];
}
// Poses::setParamsFromVector (Poses.h:72-76): pose 0 is not a parameter
__device__ __forceinline__ void set_params_lds(int n, const double* p, double* rel_o, double* rel_t) {
    for (int i = threadIdx.x; i < 3 * (n - 1); i += kWave) rel_o[3 + i] = p[i], rel_t[3 + i] = p[3 * (n - 1) + i];
    __syncthreads();
}
__device__ __forceinline__ void store_ctrl(double* ctrl, int n, const double* glob_o, const double* glob_t) {
    for (int i = threadIdx.x; i < 3 * n; i += kWave) {
        const int k = i / 3, c = i - 3 * k;
        ctrl[6 * k + c] = glob_o[i], ctrl[6 * k + 3 + c] = glob_t[i];
    }
}

struct ChainLds {
    double *rel_o, *rel_t, *glob_o, *glob_t, *E, *R, *par, *org, *rows, *dense;
};
__device__ __forceinline__ ChainLds carve(double* sm, int n, int P) {
    ChainLds c;
    c.rel_o = sm, c.rel_t = sm + 3 * n, c.glob_o = sm + 6 * n, c.glob_t = sm + 9 * n, c.E = sm + 12 * n, c.R = sm + 21 * n;
    c.par = sm + 30 * n, c.org = c.par + P, c.rows = c.org + P;
    c.dense = sm + 12 * n;  // the window's IMU rows: twelve dense-trajectory values per row in E (9 n doubles, free between two chains) and the head of R
    return c;
}
size_t chain_lds_bytes(const LoopModel& m) { return (30 * (size_t)m.n + 2 * (size_t)m.P + (size_t)(m.extra > 0 ? m.extra : 1)) * sizeof(double); }

// iteration start (:72-75) on the chain held in LDS: paramVec = getPoseParameters(); updateGlobalPoints re-chains the window model (the
// keyframe model did in setPoseParameters); ctrl0 = the global poses of the base table
__device__ void begin_body(const LoopModel& m, const ChainLds& c, double* __restrict__ state0, double* __restrict__ paramVec, double* __restrict__ ctrl0) {
    for (int i = threadIdx.x; i < 3 * (m.n - 1); i += kWave) paramVec[i] = c.rel_o[3 + i], paramVec[3 * (m.n - 1) + i] = c.rel_t[3 + i];
    if (m.model == 1) {
        wave_relative_to_global(m.n, c.rel_o, c.rel_t, c.glob_o, c.glob_t, c.E, c.R);
        store_state(state0, m.n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
    }
    store_ctrl(ctrl0, m.n, c.glob_o, c.glob_t);
}
// `state_ready` (optional): counter the Jacobian chains on the side stream wait for (dev_sync.h) -- signalled on every path, a stopped
// loop included
__global__ __launch_bounds__(kWave) void k_loop_begin(const LoopModel m, double* __restrict__ state0, double* __restrict__ paramVec, double* __restrict__ ctrl0,
                                                      LoopFlags* __restrict__ flags, uint32_t* state_ready) {
    extern __shared__ double sm[];
    DevSync sy;
    sy.signal_counter = state_ready;
    if (flags->stop == 0) {
        if (threadIdx.x == 0) flags->nan = 0;
        const ChainLds c = carve(sm, m.n, m.P);
        load_state(state0, m.n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
        begin_body(m, c, state0, paramVec, ctrl0);
    }
    dev_sync_leave(sy);
}

// One workgroup (= one wave) per evaluation of the batch.  Evaluations of a batch depend on each other only through relative pose 0 of
// the window model with IMU rows (pose0_orbit); everything else an evaluation reads is the incoming state and its own parameters.
__global__ __launch_bounds__(kWave) void k_loop_chain(const LoopModel m, int mode, const double* __restrict__ state_in, double* __restrict__ state_out,
                                                      const double* __restrict__ paramVec, const double* __restrict__ step, double increment,
                                                      double* __restrict__ ctrl, double* __restrict__ extra, const LoopFlags* __restrict__ flags, int part,
                                                      uint32_t* start_signal) {
    // part 0: everything.  The nine trials of a window with IMU rows are launched twice (optimize_loop.cpp): part 1 -- the control poses only,
    // on the main stream, where the pose tables and the trial batch wait for them -- and part 2 -- the additional rows and the state the
    // last trial leaves, on the side stream beside the trial batch (the rows are half of the chain kernel's time and nobody reads them before
    // the squared sums).  start_signal: raised as soon as the kernel runs, i.e. once everything in front of it on its stream (the LM step)
    // is done -- the side stream's part 2 waits for it.
    extern __shared__ double sm[];
    if (start_signal != nullptr && blockIdx.x == 0 && threadIdx.x == 0) dev_sync_signal(start_signal);
    if (flags->stop != 0 || (mode == 1 && flags->nan != 0)) return;
    const int n = m.n, P = m.P, a = m.extra > 0 ? m.extra : 0;
    const int b = blockIdx.x, B = gridDim.x;
    const ChainLds c = carve(sm, n, P);
    load_state(state_in, n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
    const bool carries = m.model == 1 && a > 0;  // updateImuError's global2relative rewrites relative pose 0 in every evaluation
    const Pose0 p_in{col3(c.rel_o, 0), col3(c.rel_t, 0)};
    double* my_ctrl = ctrl + (size_t)b * n * 6;
    double* my_rows = extra + (size_t)b * a;
    bool evaluate;
    if (mode == 0) {
        // evaluation 0 (:99) is the chain as loop_begin left it
        if (b == 0) store_ctrl(my_ctrl, n, c.glob_o, c.glob_t);
        // its additional rows; the window model's round trip also decides `origin` (:204 reads the parameters AFTER it), so every
        // evaluation of that model repeats it
        if (b == 0 || carries) {
            wave_extra_rows(m, c.rel_o, c.rel_t, c.glob_o, c.glob_t, c.rows, c.dense);
            if (b == 0)
                for (int i = threadIdx.x; i < a; i += kWave) my_rows[i] = c.rows[i];
        }
        // origin = getPoseParameters(); loop = origin, loop[k] += h for evaluation k + 1 (:209-212)
        for (int i = threadIdx.x; i < 3 * (n - 1); i += kWave) {
            const double o = c.rel_o[3 + i], t = c.rel_t[3 + i];
            c.org[i] = o, c.org[3 * (n - 1) + i] = t;
            c.par[i] = o, c.par[3 * (n - 1) + i] = t;
        }
        __syncthreads();
        if (b > 0 && threadIdx.x == 0) c.par[b - 1] += increment;
        __syncthreads();
        evaluate = b > 0;
    } else {
        // trial k = b + 1: raw + 0.1 k step (:160)
        for (int i = threadIdx.x; i < P; i += kWave) c.par[i] = paramVec[i] + 0.1 * (double)(b + 1) * step[i];
        __syncthreads();
        evaluate = true;
    }
    if (evaluate) {
        if (carries) {  // relative pose 0 as the b evaluations of this batch that ran before this one left it
            if (threadIdx.x == 0) {
                const Pose0 p = pose0_orbit(p_in, b);
                set_col3(c.rel_o, 0, p.o), set_col3(c.rel_t, 0, p.t);
            }
            __syncthreads();
        }
        set_params_lds(n, c.par, c.rel_o, c.rel_t);
        wave_relative_to_global(n, c.rel_o, c.rel_t, c.glob_o, c.glob_t, c.E, c.R);
        if (part != 2) store_ctrl(my_ctrl, n, c.glob_o, c.glob_t);
        if (part != 1) {
            wave_extra_rows(m, c.rel_o, c.rel_t, c.glob_o, c.glob_t, c.rows, c.dense);
            for (int i = threadIdx.x; i < a; i += kWave) my_rows[i] = c.rows[i];
        }
    }
    if (b == B - 1 && part != 1) {
        // the state the serial loop leaves behind: the last evaluation of the batch; the Jacobian batch then restores the parameters (:231)
        if (mode == 0) set_params_lds(n, c.org, c.rel_o, c.rel_t);
        store_state(state_out, n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
    }
}

__global__ __launch_bounds__(256) void k_loop_scatter_extra(const double* __restrict__ extra, int B, int a, double* __restrict__ E, int64_t ldE, int M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * a) return;
    const int b = i / a, r = i - b * a;
    E[(size_t)b * ldE + M + r] = extra[i];
}

// Gauss-Jordan with partial pivoting on [A | I] in LDS, the element operations of host_math.cpp's lm_solve (serial branch) one to one.
// Lane = column of the augmented matrix (row-major in LDS: a row is a conflict-free access), wave w owns the rows r = w (mod kSolveWaves).
// A pivot step is one round of LDS reads (pivot column for the search and as multipliers, the two rows that swap, the wave's own rows),
// one barrier, arithmetic in registers, one round of writes, one barrier: the search and the pivot row are computed redundantly by
// every wave, so nothing is exchanged between waves but the matrix itself.  Multipliers reach the lanes through v_readlane (the lane
// that read M[r][c0] is lane r).  Columns of A left of the pivot hold finished entries nobody reads again; they are updated along.
constexpr int kSolveWaves = 8, kSolveRows = kLoopSolveMaxP / kSolveWaves;  // rows per wave (upper bound)
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// max over the wave of values >= 0 (or the sentinel -1) without the LDS crossbar: DPP row shifts and row broadcasts (an inclusive
// max-scan whose last lane holds the maximum); lanes a step does not reach see 0, which never exceeds a real maximum
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kCtrl, kRowMask, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kCtrl, kRowMask, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_nonneg(double v) {
    double o;
    o = dpp_mov_f64<0x111, 0xf>(v), v = o > v ? o : v;  // row_shr:1
    o = dpp_mov_f64<0x112, 0xf>(v), v = o > v ? o : v;  // row_shr:2
    o = dpp_mov_f64<0x114, 0xf>(v), v = o > v ? o : v;  // row_shr:4
    o = dpp_mov_f64<0x118, 0xf>(v), v = o > v ? o : v;  // row_shr:8
    o = dpp_mov_f64<0x142, 0xa>(v), v = o > v ? o : v;  // row_bcast:15 -> rows 1 and 3
    o = dpp_mov_f64<0x143, 0xc>(v), v = o > v ? o : v;  // row_bcast:31 -> rows 2 and 3
    return readlane_f64(v, 63);
}
// element (i, j) of Hp = [J | e0]^T [J | e0]: either reduced already, or the block sums of the normal-equation kernel added here in block
// order (the loads of sixteen blocks in flight together, only the adds are a chain) -- one dispatch less between the batch and the step
struct HpSource {
    const double* Hp;       // (P+1)^2 column-major, or null
    const double* partial;  // block sums (dmsa_kernels.h: NormalEqPartials)
    int nsplit, nt;
};
__device__ __forceinline__ double hp_element(const HpSource& h, int n1, int i, int j) {
    if (h.Hp) return h.Hp[(size_t)j * n1 + i];
    const double* src = h.partial + ((size_t)(j >> 5) * h.nt + (i >> 5)) * 1024 + (j & 31) * 32 + (i & 31);
    const size_t stride = (size_t)h.nt * h.nt * 1024;
    double s = 0.0;
    int sp = 0;
    for (; sp + 32 <= h.nsplit; sp += 32) {
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = src[(size_t)(sp + u) * stride];
#pragma unroll
        for (int u = 0; u < 32; ++u) s += v[u];
    }
    if (sp < h.nsplit) {  // the tail in one batch as well
        double v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = sp + u < h.nsplit ? src[(size_t)(sp + u) * stride] : 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (sp + u < h.nsplit) s += v[u];
    }
    return s;
}
template <int kColsPerLane, int kRows /* rows per wave: P <= kRows * kSolveWaves */>
__global__ __launch_bounds__(kSolveWaves* kWave) void k_loop_lm_step(const HpSource hp, int P, double lambda, double alpha, double max_step,
                                                                      double* __restrict__ step, LoopFlags* __restrict__ flags, double* __restrict__ error0_out) {
    extern __shared__ double sm[];
    if (flags->stop != 0) return;
    const int W = 2 * P, lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, n1 = P + 1;
    double* M = sm;                    // P x W row-major: [A | inv]
    double* g = sm + (size_t)P * W;    // P
    // [H + lambda I | g] and e0^T e0: the (P+1)^2 - 1 - P elements that matter spread evenly over the threads (every element may be a sum
    // of block sums: as many of those loads in flight as possible)
    for (int idx = threadIdx.x; idx < n1 * n1; idx += kSolveWaves * kWave) {
        const int j = idx / n1, i = idx - j * n1;  // element (i, j), column-major like Hp
        if (j < P && i < P) {
            double v = hp_element(hp, n1, i, j);  // H(i, j), damped on the diagonal (:110)
            if (i == j) v += lambda;
            M[(size_t)i * W + j] = v;
        } else if (j == P) {
            const double v = hp_element(hp, n1, i, P);  // g = J^T e0 (last column), e0^T e0 for the end of the iteration
            if (i < P)
                g[i] = v;
            else if (error0_out)
                *error0_out = v;
        }
    }
    for (int idx = threadIdx.x; idx < P * P; idx += kSolveWaves * kWave) {
        const int r = idx / P, c = idx - r * P;
        M[(size_t)r * W + P + c] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int c0 = 0; c0 < P; ++c0) {
        // ---- reads of the state before the step ----
        const bool in = lane >= c0 && lane < P;
        const double fcol = lane < P ? M[(size_t)lane * W + c0] : 0.0;  // lane = row: the pivot column
        double x[kRows][kColsPerLane];
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int r = wave + j * kSolveWaves;
#pragma unroll
            for (int q = 0; q < kColsPerLane; ++q) {
                const int c = lane + q * kWave;
                x[j][q] = (r < P && c < W) ? M[(size_t)r * W + c] : 0.0;
            }
        }
        // first strict maximum of |A[r][c0]| over r >= c0 (a NaN never wins; a NaN on the diagonal keeps the diagonal)
        double v = in ? fabs(fcol) : -1.0;
        if (isnan(v)) v = -1.0;
        const double mx = wave_max_nonneg(v);
        const unsigned long long hit = __ballot(in && v == mx);
        const double diag = readlane_f64(fcol, c0);
        const int piv = isnan(diag) || hit == 0ull ? c0 : (int)(__ffsll((long long)hit) - 1);
        const double d = readlane_f64(fcol, piv);    // the pivot
        const double f_swapped = diag;               // multiplier of the row that moves from c0 to piv
        double top[kColsPerLane], srow[kColsPerLane];
#pragma unroll
        for (int q = 0; q < kColsPerLane; ++q) {
            const int c = lane + q * kWave;
            top[q] = c < W ? M[(size_t)c0 * W + c] : 0.0;
            const double pv = c < W ? M[(size_t)piv * W + c] : 0.0;
            srow[q] = pv / d;  // the scaled pivot row, which becomes row c0
        }
        __syncthreads();  // every read of the old state is done
        // ---- writes: row c0 <- scaled pivot row; row piv <- old row c0, updated; all other rows updated ----
#pragma unroll
        for (int j = 0; j < kRows; ++j) {
            const int r = wave + j * kSolveWaves;
            if (r >= P) continue;
            const bool is_top = r == c0, is_swapped = r == piv && piv != c0;
            const double f = is_swapped ? f_swapped : readlane_f64(fcol, r);
#pragma unroll
            for (int q = 0; q < kColsPerLane; ++q) {
                const int c = lane + q * kWave;
                if (c >= W) continue;
                const double old = is_swapped ? top[q] : x[j][q];
                const double upd = f == 0.0 ? old : old - f * srow[q];
                M[(size_t)r * W + c] = is_top ? srow[q] : upd;
            }
        }
        __syncthreads();
    }
    // :113 step = (-alpha H^-1) g, row by row (lane = row; P <= 64), then the NaN test and the clamp of :116-128 in registers
    if (wave != 0) return;
    double s = 0.0;
    if (lane < P)
        for (int j = 0; j < P; ++j) s += (-alpha * M[(size_t)lane * W + P + j]) * g[j];
    if (__ballot(lane < P && isnan(s)) != 0ull) {
        if (lane == 0) flags->nan = 1;
        return;
    }
    double mx = lane < P ? s : -INFINITY, mn = lane < P ? s : INFINITY;  // max / min do not depend on the order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double a = __shfl_xor(mx, o), c = __shfl_xor(mn, o);
        mx = mx < a ? a : mx;
        mn = c < mn ? c : mn;
    }
    const double neg = -mn;
    const double max_elem = mx < neg ? neg : mx;  // std::max(mx, -mn)
    if (max_elem > max_step) s = (max_step / max_elem) * s;
    if (lane < P) step[lane] = s;
}

// ---- P > 64: the same Gauss-Jordan inverse, blocked over column blocks of the augmented matrix [A | I], one workgroup per block ------
// Every element sees the operations of the serial algorithm in step order -- pivot row: x / d_k, other rows: x - f_rk * s_kc (skipped
// for f_rk == 0) -- so the result is the serial one bit for bit (as host_math.cpp's blocked lm_solve argues).  What a step needs, the
// pivot choice, d_k and the multipliers f_rk, depends only on column k; K = 8 consecutive pivot columns form a PANEL whose evolution
// depends on nothing but itself.  Workgroup b owns columns [8b, 8b + 8) of all rows (thread = row, the 8 entries in registers) and, panel
// by panel, factors a private copy of the panel in lockstep with its own block: per step one argmax over the rows, one division of
// the pivot row's 16 entries by the thread that owns it, one multiply-subtract of 16 entries by everybody else.  The only traffic
// between workgroups: the owner of the next panel publishes its block (n x 8 doubles) when it has applied the current panel, and the
// others pick it up -- a chain of P / 8 hand-overs instead of P, no grid-wide barrier.  A-blocks retire once they have served as
// panel; the last workgroup to finish multiplies the inverse with g and clamps the step (:113-128).
#ifndef DMSA_PANEL
#define DMSA_PANEL 8
#endif
constexpr int kPanel = DMSA_PANEL;  // pivot columns per panel = columns per workgroup
constexpr int kPanelLdsHead = 2 * kPanel + 8 + 16 + 16;  // doubles in front of s_perm: pivot row (+ pivot), per-wave best value / row
struct PanelSolveWork {   // device scratch of one solve (sized by loop_panel_solve_bytes)
    unsigned int epoch_flags_offset;  // unused placeholder: layout is computed from P
};
__device__ __forceinline__ void agent_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double agent_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_old_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), kCtrl, kRowMask, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), kCtrl, kRowMask, 0xf, false);
    return __hiloint2double(hi, lo);
}
// (value, logical row) of the wave's best candidate: largest value, among equals the smallest logical row; value -1 = no candidate.
// The maximum runs through a DPP max-scan; its owner is found with one ballot -- ties (equal |x| in two rows) are rare and walked bit by bit.
__device__ __forceinline__ void wave_best(double& v, int& l) {
    double m = v, o;
    o = dpp_old_f64<0x111, 0xf>(-2.0, m), m = o > m ? o : m;  // row_shr:1
    o = dpp_old_f64<0x112, 0xf>(-2.0, m), m = o > m ? o : m;  // row_shr:2
    o = dpp_old_f64<0x114, 0xf>(-2.0, m), m = o > m ? o : m;  // row_shr:4
    o = dpp_old_f64<0x118, 0xf>(-2.0, m), m = o > m ? o : m;  // row_shr:8
    o = dpp_old_f64<0x142, 0xa>(-2.0, m), m = o > m ? o : m;  // row_bcast:15
    o = dpp_old_f64<0x143, 0xc>(-2.0, m), m = o > m ? o : m;  // row_bcast:31
    const double mx = readlane_f64(m, 63);
    if (mx < 0.0) {  // no candidate in this wave
        v = -1.0, l = 0x7fffffff;
        return;
    }
    unsigned long long hit = __ballot(v == mx);
    int best = __builtin_amdgcn_readlane(l, __builtin_ctzll(hit));
    hit &= hit - 1ull;
    while (hit != 0ull) {  // another lane with the same value: the smaller logical row wins
        const int other = __builtin_amdgcn_readlane(l, __builtin_ctzll(hit));
        best = other < best ? other : best;
        hit &= hit - 1ull;
    }
    v = mx, l = best;
}
__global__ __launch_bounds__(1024) void k_loop_lm_panels(const double* __restrict__ Hp, int P, double lambda, double alpha, double max_step,
                                                         double* __restrict__ work, unsigned int epoch, double* __restrict__ step, LoopFlags* __restrict__ flags) {
    if (flags->stop != 0) return;
    const int n = P, n1 = P + 1, r = threadIdx.x, b = blockIdx.x, nblocks = gridDim.x;
    const int npanels = (n + kPanel - 1) / kPanel;
    double* pub = work;                                            // [npanels][n][8]
    double* inv_out = pub + (size_t)npanels * n * kPanel;          // [n][n] physical rows
    unsigned long long* ready = reinterpret_cast<unsigned long long*>(inv_out + (size_t)n * n);  // [npanels]: epoch when published
    unsigned long long* done = ready + npanels;                    // [0]: (epoch << 32) | finished workgroups
    int* iperm_out = reinterpret_cast<int*>(done + 2);             // [n]: logical row of every physical row after the last step
    extern __shared__ double sm[];
    double* s_prow = sm;                        // 16 (+1): pivot row entries (panel | own block), raw then scaled; [16] = the pivot
    double* s_bv = sm + 2 * kPanel + 8;         // per wave: best value (<= 16 waves)
    int* s_bl = reinterpret_cast<int*>(sm + 2 * kPanel + 8 + 16);   // per wave: best logical row
    int* s_perm = s_bl + 16;                    // logical -> physical
    __shared__ int s_last;
    const bool row = r < n;
    const int lane = r & 63, wave = r >> 6, nwaves = (blockDim.x + 63) >> 6;
    // own block: columns c0 .. c0 + 7 of [A | I]
    const int c0 = b * kPanel;
    double xo[kPanel];
#pragma unroll
    for (int c = 0; c < kPanel; ++c) {
        const int col = c0 + c;
        double v = 0.0;
        if (row) {
            if (col < n) {
                v = Hp[(size_t)col * n1 + r];  // H(r, col), damped on the diagonal (:110)
                if (col == r) v += lambda;
            } else if (col < 2 * n) {
                v = (col - n) == r ? 1.0 : 0.0;
            }
        }
        xo[c] = v;
    }
    for (int i = r; i < n; i += blockDim.x) s_perm[i] = i;
    int logical = r;      // iperm[r]
    bool pivoted = false;
    if (b == 0 && row) {  // block 0 is panel 0 as it stands
#pragma unroll
        for (int c = 0; c < kPanel; ++c) pub[(size_t)r * kPanel + c] = xo[c];
    }
    if (b == 0) __threadfence();
    __syncthreads();
    if (b == 0 && r == 0) __hip_atomic_store(ready, (unsigned long long)epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const int last_panel_of_block = c0 + kPanel <= n ? b : npanels - 1;  // an A block is dead once it has been the panel
#ifdef DMSA_PANEL_TIMING
    long long t_wait = 0, t_load = 0, t_steps = 0, t_pub = 0, t_begin = wall_clock64();
#endif
    for (int p = 0; p <= last_panel_of_block && p < npanels; ++p) {
#ifdef DMSA_PANEL_TIMING
        const long long ta = wall_clock64();
#endif
        const int k0 = p * kPanel, kp = min(kPanel, n - k0);
        const bool is_panel = p == b;
        // ---- the panel as its owner published it ----
        // Hand-over: the owner's plain stores are released by its fence + flag; consumers poll the flag with a relaxed load (an acquire
        // load would invalidate the L2 on every poll) and fence once after they have seen it.  A hand-over costs ~3 us (-DDMSA_PANEL_TIMING
        // prints where a block's time goes); write-through stores / agent-scope atomic loads instead of the fences were no faster.  What
        // bounds the solve is the pivot step itself: ~1.2 us of dependent LDS round trips, wave reductions and one fp64 division.
        if (r == 0)
            while (__hip_atomic_load(ready + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)epoch) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
#ifdef DMSA_PANEL_TIMING
        const long long tb = wall_clock64();
#endif
        __threadfence();
        double xp[kPanel];
        {
            const double2* src = reinterpret_cast<const double2*>(pub + ((size_t)p * n + (row ? r : 0)) * kPanel);
#pragma unroll
            for (int c = 0; c < kPanel / 2; ++c) {
                const double2 v = src[c];
                xp[2 * c] = row ? v.x : 0.0, xp[2 * c + 1] = row ? v.y : 0.0;
            }
        }
#ifdef DMSA_PANEL_TIMING
        __syncthreads();
        const long long tc = wall_clock64();
#endif
        // ---- kp pivot steps on the panel copy and on the own block in lockstep ----
#pragma unroll
        for (int j = 0; j < kPanel; ++j) {
            if (j >= kp) break;
            const int k = k0 + j;
            // pivot: largest |x| among the rows not yet pivoted, ties to the smallest logical row (= the serial search from row k down)
            double v = row && !pivoted ? fabs(xp[j]) : -1.0;
            if (isnan(v)) v = -1.0;
            int l = row && !pivoted ? logical : 0x7fffffff;
            wave_best(v, l);
            if (lane == 0) s_bv[wave] = v, s_bl[wave] = l;
            __syncthreads();
            // the waves' candidates meet in every wave: lane w takes wave w's, one more reduction (no chain of LDS reads)
            double bv = lane < nwaves ? s_bv[lane] : -1.0;
            int bl = lane < nwaves ? s_bl[lane] : 0x7fffffff;
            wave_best(bv, bl);
            if (bv < 0.0) bl = k;  // nothing comparable (NaNs): keep the diagonal like the serial search
            const int pp = s_perm[bl], p0 = s_perm[k];
            const bool mine = row && r == pp;
            // the pivot row's 16 entries are divided by 16 lanes (one fp64 division is ~40 dependent instructions; sixteen of them on
            // the one lane that owns the row would dominate the step)
            if (mine) {
#pragma unroll
                for (int c = 0; c < kPanel; ++c) s_prow[c] = xp[c], s_prow[kPanel + c] = xo[c];
                s_prow[2 * kPanel] = xp[j];
            }
            __syncthreads();  // (also: everybody has read s_perm / s_bv / s_bl of this step)
            if (r < 2 * kPanel) {
                const double q = s_prow[r] / s_prow[2 * kPanel];
                __builtin_amdgcn_wave_barrier();  // all sixteen lanes (one wave) have read the pivot before any overwrites
                s_prow[r] = q;
            }
            if (row && r == pp) logical = k, pivoted = true;
            else if (row && r == p0) logical = bl;
            if (r == 0) s_perm[k] = pp, s_perm[bl] = p0;
            __syncthreads();
            if (row) {
                if (mine) {
#pragma unroll
                    for (int c = 0; c < kPanel; ++c) xp[c] = s_prow[c], xo[c] = s_prow[kPanel + c];
                } else {
                    const double f = xp[j];
                    if (f != 0.0) {
#pragma unroll
                        for (int c = 0; c < kPanel; ++c) {  // (16-byte reads of the pivot row measured slower: 443 vs 405 us per solve at P = 186)
                            xp[c] -= f * s_prow[c];
                            xo[c] -= f * s_prow[kPanel + c];
                        }
                    }
                }
            }
            // no barrier here: the next step's first barrier separates these reads of s_prow from its next writer
        }
#ifdef DMSA_PANEL_TIMING
        __syncthreads();
        const long long td = wall_clock64();
#endif
        if (is_panel) {
            // the block was its own panel: what happened to the copy happened to the block
#pragma unroll
            for (int c = 0; c < kPanel; ++c) xo[c] = xp[c];
        }
        // ---- hand the next panel over as soon as it exists ----
        if (b == p + 1 && p + 1 < npanels) {
            if (row) {
                double2* dst = reinterpret_cast<double2*>(pub + ((size_t)(p + 1) * n + r) * kPanel);
#pragma unroll
                for (int c = 0; c < kPanel / 2; ++c) dst[c] = double2{xo[2 * c], xo[2 * c + 1]};
            }
            __threadfence();
            __syncthreads();
            if (r == 0) __hip_atomic_store(ready + p + 1, (unsigned long long)epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef DMSA_PANEL_TIMING
        t_wait += tb - ta, t_load += tc - tb, t_steps += td - tc, t_pub += wall_clock64() - td;
#endif
    }
#ifdef DMSA_PANEL_TIMING
    if (r == 0 && (b == nblocks - 1 || b == 12 || b == 13))
        printf("[panel timing] block %d: total %lld  wait %lld  fence+load %lld  steps %lld  publish %lld (x10 ns), %d panels\n", b, wall_clock64() - t_begin, t_wait,
               t_load, t_steps, t_pub, min(last_panel_of_block + 1, npanels));
#endif
    // ---- columns of the inverse ----
    if (row)
#pragma unroll
        for (int c = 0; c < kPanel; ++c) {
            const int col = c0 + c;
            if (col >= n && col < 2 * n) inv_out[(size_t)r * n + (col - n)] = xo[c];
        }
    // the last block always lives to the last panel: its row permutation is the final one
    if (b == nblocks - 1 && row) iperm_out[r] = logical;
    __threadfence();
    __syncthreads();
    if (r == 0) {
        // count finished workgroups of THIS solve (the word carries the epoch; a stale epoch restarts the count)
        unsigned long long seen = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want;
        do {
            want = (seen >> 32) == (unsigned long long)epoch ? seen + 1ull : (((unsigned long long)epoch << 32) | 1ull);
        } while (!__hip_atomic_compare_exchange_strong(done, &seen, want, __ATOMIC_ACQ_REL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        s_last = (int)(want & 0xffffffffull) == nblocks;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // :113 step = (-alpha H^-1) g, row by row; the physical row r is logical row iperm[r] (a retired A block stopped tracking the
    // permutation, so it is read back from the block that certainly saw every step)
    double* s_step = sm + kPanelLdsHead + ((size_t)n + 1) / 2 + 1;   // behind s_perm (ints)
    double sres = 0.0;
    if (row) {
        // the row of the inverse arrives eight entries at a time (independent loads in flight), the sum itself is a chain in column order
        const double* irow = inv_out + (size_t)r * n;
        const double* gv = Hp + (size_t)P * n1;
        int j = 0;
        for (; j + 8 <= n; j += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = irow[j + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) sres += (-alpha * v[u]) * gv[j + u];
        }
        for (; j < n; ++j) sres += (-alpha * irow[j]) * gv[j];
        s_step[iperm_out[r]] = sres;
    }
    __syncthreads();
    // NaN test and clamp (:116-128): wave 0 walks the step (max / min do not depend on the order)
    if (wave == 0) {
        bool any_nan = false;
        double mx = -INFINITY, mn = INFINITY;
        for (int i = lane; i < n; i += 64) {
            const double v = s_step[i];
            any_nan = any_nan || isnan(v);
            mx = mx < v ? v : mx;
            mn = v < mn ? v : mn;
        }
        if (__ballot(any_nan) != 0ull) {
            if (lane == 0) flags->nan = 1;
        } else {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double a = __shfl_xor(mx, o), c = __shfl_xor(mn, o);
                mx = mx < a ? a : mx;
                mn = c < mn ? c : mn;
            }
            const double neg = -mn;
            const double max_elem = mx < neg ? neg : mx;
            for (int i = lane; i < n; i += 64) {
                const double v = s_step[i];
                step[i] = max_elem > max_step ? (max_step / max_elem) * v : v;
            }
        }
    }
}

// ---- 64 < P <= 192: the same Gauss-Jordan inverse as a STREAM of pivot steps (round 4) ---------------------------------------------------
// k_loop_lm_panels pays, per panel of 8 pivot steps, one hand-over through global memory (~3 us) and three workgroup barriers per step
// (~1.3 us): 24 x (3 + 8 x 1.3) = 310 us at P = 186, all of it on the critical path.  Here a pivot step is the work of ONE wave and the
// critical path never leaves the compute unit it is on for 64 steps at a time:
//   * a WORKER wave owns 8 columns of [A | I] for all rows (lane = row mod 64, R rows per lane, everything in registers) -- no workgroup
//     barrier anywhere, what the lanes of a wave exchange goes through a few LDS words in program order;
//   * what a pivot step needs from the outside is its RECORD: the physical pivot row and, for every row r, the entry x(r, k) as it was
//     before the step (the multiplier f_rk; for the pivot row that is the pivot d_k).  The worker that owns column k makes the record
//     (pivot search in its own registers: the column is up to date because it has applied every earlier record), every other worker
//     only applies records: pivot row of its own columns / d_k, x - f_rk * s_kc for the rest -- the element operations of the serial
//     algorithm in step order, bit for bit (skipped where f_rk == 0, like there);
//   * records travel through a ring in LDS to the other workers of the workgroup (eight workers = 64 columns: the next owner is one
//     LDS round trip behind, not one global hand-over) and, copied out by a HELPER wave, through global memory to the other workgroups
//     (`published` counts them; the helper of a receiving workgroup copies them into its own ring).  Workgroups of A columns take over
//     the factorisation one after the other (two global hand-overs at P = 186 instead of 23); the workgroups of I columns only follow;
//   * A columns left of the pivot are dead (unit vectors nobody reads again): a worker retires when its last column has been the pivot.
// The inverse leaves the kernel transposed (inv_t[c][r], coalesced for the product with g that k_loop_lm_stream_tail computes row by
// row in column order).  Deadlock freedom as for the panels: a workgroup only waits for workgroups with smaller block indices, which
// the dispatcher places first, and inside a workgroup for waves that never wait for it.
#ifndef DMSA_STREAM_W
#define DMSA_STREAM_W 8
#endif
#ifndef DMSA_STREAM_WORKERS
#define DMSA_STREAM_WORKERS 8
#endif
constexpr int kStreamW = DMSA_STREAM_W;              // columns per worker wave (8 or 16)
constexpr int kStreamWorkers = DMSA_STREAM_WORKERS;  // worker waves per workgroup (+ one helper wave); <= 15
#ifndef DMSA_STREAM_D
#define DMSA_STREAM_D 24
#endif
#ifndef DMSA_STREAM_BATCH
#define DMSA_STREAM_BATCH 8
#endif
constexpr int kStreamD = DMSA_STREAM_D;          // ring slots (pivot steps a consumer may lag behind its producer)
constexpr int kStreamBatch = DMSA_STREAM_BATCH;  // records a helper moves per round; the producer checks the ring for room every kStreamBatch steps
constexpr int kStreamNever = 0x7fffffff;
template <int R>
struct StreamLds {
    double ring_f[kStreamD][64 * R];                 // records: x(r, k) before step k (f_rk; d_k in the pivot row's place)
    double scr[kStreamWorkers][2 * kStreamW];        // per worker: its pivot row's entries, raw | scaled
    int ring_pp[kStreamD];                           // records: physical pivot row
    int produced;                                    // records [.., produced) are in the ring
    int consumed[16];                                // per consumer (workers, then the helper): records it needs no more; kStreamNever = none
};
struct StreamWork {   // carved from the solve scratch (loop_panel_solve_doubles covers both layouts)
    double* mult;                  // [P][64 R]
    double* inv_t;                 // [P][64 R]: column c of the inverse, physical rows
    int* gpp;                      // [P]
    int* iperm;                    // [64 R]: logical row of every physical row after the last step
    unsigned long long* published; // (epoch << 32) | records published
};
__device__ __host__ __forceinline__ StreamWork stream_work(double* work, int P, int RN) {
    StreamWork w;
    w.mult = work;
    w.inv_t = work + (size_t)P * RN;
    w.published = reinterpret_cast<unsigned long long*>(w.inv_t + (size_t)P * RN);
    w.gpp = reinterpret_cast<int*>(w.published + 2);
    w.iperm = w.gpp + ((P + 1) / 2) * 2;
    return w;
}
// every wait of the stream kernel is for a wave that never waits for the waiter (records only flow forward; workgroups with smaller block
// indices are placed first), so a wait that does not end is a bug: after ~seconds of polling the wave traps -- the dispatch dies with an
// error instead of hanging the device
constexpr int kStreamSpinLimit = 1 << 25;
__device__ __forceinline__ void stream_spin(int& spins) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kStreamSpinLimit) __builtin_trap();
}
__device__ __forceinline__ int lds_load_acquire(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_release(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// A count that follows LDS WRITES of the same wave needs no wait: the LDS takes a wave's instructions in order, so whoever sees the count
// sees the writes issued before it.  (A release store would first wait for those writes to come back: ~100 cycles on the critical path
// of every pivot step.)  Only the compiler has to be kept from moving the writes below the count.
__device__ __forceinline__ void lds_store_after_writes(int* p, int v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}
// smallest entry of consumed[0..15] (every lane returns it)
__device__ __forceinline__ int stream_min_consumed(const int* consumed, int lane) {
    int v = lds_load_acquire(consumed + (lane & 15));
    v = min(v, __shfl_xor(v, 1));
    v = min(v, __shfl_xor(v, 2));
    v = min(v, __shfl_xor(v, 4));
    v = min(v, __shfl_xor(v, 8));
    return __builtin_amdgcn_readfirstlane(v);
}
// Which of a lane's R rows a (wave-uniform) slot number means, as R separate flags the optimiser cannot see through: written as
// `ps == i` inside the unrolled loops below, the compiler folds the if-chains into x[ps][c] -- a dynamically indexed array, i.e. the
// whole register block of a worker in scratch memory.
template <int R>
struct SlotFlags {
    bool is[R];
};
template <int R>
__device__ __forceinline__ SlotFlags<R> slot_flags(int ps) {
    SlotFlags<R> s;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        int hit = __builtin_amdgcn_readfirstlane(ps == i ? 1 : 0);
        asm volatile("" : "+s"(hit));
        s.is[i] = hit != 0;
    }
    return s;
}
// One pivot step applied to a worker's columns c >= kFirst: pivot row (lane pl, slot `sel`) -> x / d, every other row x - f * s.
template <int R, int kFirst>
__device__ __forceinline__ void stream_apply(double (&x)[R][kStreamW], const double (&f)[R], double d, int pl, const SlotFlags<R>& sel, int lane, double* scr) {
    if (lane == pl) {
#pragma unroll
        for (int i = 0; i < R; ++i)
            if (sel.is[i]) {
#pragma unroll
                for (int c = kFirst; c < kStreamW; ++c) scr[c] = x[i][c];
            }
    }
    __builtin_amdgcn_wave_barrier();
    // one lane per column divides (one fp64 division is ~40 dependent instructions)
    if (lane >= kFirst && lane < kStreamW) scr[kStreamW + lane] = scr[lane] / d;
    __builtin_amdgcn_wave_barrier();
    double q[kStreamW];
#pragma unroll
    for (int c = kFirst; c < kStreamW; ++c) q[c] = scr[kStreamW + c];
    __builtin_amdgcn_wave_barrier();   // the next step's writes to scr come after these reads
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (lane == pl && sel.is[i]) {
#pragma unroll
            for (int c = kFirst; c < kStreamW; ++c) x[i][c] = q[c];
        } else if (f[i] != 0.0) {
#pragma unroll
            for (int c = kFirst; c < kStreamW; ++c) x[i][c] -= f[i] * q[c];
        }
    }
}
// the row bookkeeping of a step: the pivot row (lane pl, slot `sel`) becomes logical row k, the row that was logical row k takes its place
template <int R>
__device__ __forceinline__ void stream_swap_rows(int (&logical)[R], unsigned& pivoted, int k, int pl, const SlotFlags<R>& sel, int lane) {
    int bl = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int li = __builtin_amdgcn_readlane(logical[i], pl);   // (scalars: nothing to index)
        bl = sel.is[i] ? li : bl;
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        if (lane == pl && sel.is[i]) logical[i] = k, pivoted |= 1u << i;
        else if (logical[i] == k) logical[i] = bl;
    }
}
// Pivot step k = local column J of the worker that owns it: search, record, then the step itself on the columns right of J.
#ifdef DMSA_STREAM_TIMING
__device__ long long g_stream_phase[4];   // clock64 ticks of all factor steps: search, record, bookkeeping, apply
#endif
// The pivot search on integer keys.  |x| of a double orders like its bit pattern, so the maximum over the wave is two 32-bit max-scans
// (high words, then the low words of the lanes that hold the largest high word), each stage ONE instruction: v_max_u32 with the DPP
// shift as a modifier of its operand.  (wave_best does the same on doubles: two DPP moves, a 64-bit compare and two selects per stage,
// every one waiting for the one before -- measured as 45 % of a pivot step of the stream solve.)  key = bits(|x|) + 1 for a candidate,
// 0 for a lane without one.
template <int kCtrl, int kRowMask>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
    const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xf, true);   // lanes the shift does not reach: 0
    return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_max_u32<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max_u32<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max_u32<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max_u32<0x118, 0xf>(v);  // row_shr:8
    v = dpp_max_u32<0x142, 0xa>(v);  // row_bcast:15 -> rows 1 and 3
    v = dpp_max_u32<0x143, 0xc>(v);  // row_bcast:31 -> rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// Every lane brings (key, logical row); returns the lane of the winner -- largest key, among equals the smallest logical row -- or -1
// when no lane has a candidate.
__device__ __forceinline__ int wave_best_key(unsigned long long key, int l) {
    const unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
    const unsigned mhi = wave_max_u32(hi);
    const unsigned mlo = wave_max_u32(hi == mhi ? lo : 0u);
    if ((mhi | mlo) == 0u) return -1;
    unsigned long long hit = __ballot(hi == mhi && lo == mlo);
    int best_lane = __builtin_ctzll(hit);
    hit &= hit - 1ull;
    if (hit != 0ull) {  // equal |x| in several rows (rare): the smallest logical row wins
        int best = __builtin_amdgcn_readlane(l, best_lane);
        while (hit != 0ull) {
            const int ln = __builtin_ctzll(hit), other = __builtin_amdgcn_readlane(l, ln);
            if (other < best) best = other, best_lane = ln;
            hit &= hit - 1ull;
        }
    }
    return best_lane;
}
template <int R, int J>
__device__ __forceinline__ void stream_factor_step(double (&x)[R][kStreamW], int (&logical)[R], unsigned& pivoted, int k, int n, int lane, StreamLds<R>& sm, double* scr) {
    int spins = 0;
#ifdef DMSA_STREAM_TIMING
    const long long tp0 = clock64();
#endif
    if (J % kStreamBatch == 0)   // room for the next kStreamBatch records: every consumer is done with the ones they overwrite
        while (stream_min_consumed(sm.consumed, lane) < min(k + kStreamBatch, n) - kStreamD) stream_spin(spins);
    // pivot: largest |x| among the rows not yet pivoted, ties to the smallest logical row (= the serial search from row k down)
    unsigned long long km = 0ull;
    int lm = kStreamNever, sl = 0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const double a = fabs(x[i][J]);
        const bool cand = ((pivoted >> i) & 1u) == 0u && !isnan(a);
        const unsigned long long key = cand ? (unsigned long long)__double_as_longlong(a) + 1ull : 0ull;
        if (key > km || (key == km && key != 0ull && logical[i] < lm)) km = key, lm = logical[i], sl = i;
    }
    int pl = wave_best_key(km, lm), ps = 0;
    if (pl >= 0) {
        ps = __builtin_amdgcn_readlane(sl, pl);
    } else {
        pl = 0;
        // nothing comparable (NaNs): keep the diagonal like the serial search
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const unsigned long long hit = __ballot(logical[i] == k);
            if (hit != 0ull) pl = __builtin_ctzll(hit), ps = i;
        }
    }
#ifdef DMSA_STREAM_TIMING
    const long long tp1 = clock64();
#endif
    // the record: the column as it stands; the pivot row's entry is the pivot
    double f[R];
#pragma unroll
    for (int i = 0; i < R; ++i) f[i] = x[i][J];
    pl = __builtin_amdgcn_readfirstlane(pl), ps = __builtin_amdgcn_readfirstlane(ps);   // (uniform already; now the compiler knows)
    const SlotFlags<R> sel = slot_flags<R>(ps);
    double d = 0.0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const double di = readlane_f64(f[i], pl);
        d = sel.is[i] ? di : d;
    }
    const int slot = k % kStreamD;
#pragma unroll
    for (int i = 0; i < R; ++i) sm.ring_f[slot][lane + 64 * i] = f[i];
    if (lane == 0) sm.ring_pp[slot] = pl + 64 * ps;
    __builtin_amdgcn_wave_barrier();
    lds_store_after_writes(&sm.produced, k + 1);
#ifdef DMSA_STREAM_TIMING
    const long long tp2 = clock64();
#endif
    stream_swap_rows<R>(logical, pivoted, k, pl, sel, lane);
#ifdef DMSA_STREAM_TIMING
    const long long tp3 = clock64();
#endif
    // columns <= J of this worker are dead after the step (unit vectors nobody reads again)
    stream_apply<R, (J + 1 < kStreamW ? J + 1 : kStreamW)>(x, f, d, pl, sel, lane, scr);
#ifdef DMSA_STREAM_TIMING
    if (lane == 0) {
        const long long tp4 = clock64();
        atomicAdd((unsigned long long*)&g_stream_phase[0], (unsigned long long)(tp1 - tp0)), atomicAdd((unsigned long long*)&g_stream_phase[1], (unsigned long long)(tp2 - tp1));
        atomicAdd((unsigned long long*)&g_stream_phase[2], (unsigned long long)(tp3 - tp2)), atomicAdd((unsigned long long*)&g_stream_phase[3], (unsigned long long)(tp4 - tp3));
    }
#endif
}
template <int R>
__global__ __launch_bounds__((kStreamWorkers + 1) * kWave) void k_loop_lm_stream(const double* __restrict__ Hp, int P, double lambda, double* __restrict__ work,
                                                                                  unsigned int epoch, const LoopFlags* __restrict__ flags) {
    if (flags->stop != 0) return;
    constexpr int RN = 64 * R, W = kStreamW;
    const int n = P, n1 = P + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.x;
    const int nA = (n + W - 1) / W;                                   // worker waves of A columns (and as many of I columns)
    const int GA = (nA + kStreamWorkers - 1) / kStreamWorkers;        // workgroups of A columns
    const bool a_part = b < GA;
    const StreamWork gw = stream_work(work, P, RN);
    __shared__ StreamLds<R> sm;
    // records this workgroup makes itself: [s0, s1); everything before s0 arrives through global memory
    const int s0 = a_part ? min(n, b * kStreamWorkers * W) : n;
    const int s1 = a_part ? min(n, s0 + kStreamWorkers * W) : n;
    const int wi = (a_part ? b : b - GA) * kStreamWorkers + wave;     // worker index within its part
    const int c0 = wi * W;                                            // first column (of A or of I)
    const bool worker = wave < kStreamWorkers && c0 < n;
    if (threadIdx.x == 0) sm.produced = 0;
    if (threadIdx.x < 16) {
        // workers start at record 0 (set HERE, before anybody may write the ring), the helper reads no record before s0
        const int t = threadIdx.x;
        const bool valid = t < kStreamWorkers && ((a_part ? b : b - GA) * kStreamWorkers + t) * W < n;
        sm.consumed[t] = t == kStreamWorkers ? s0 : valid ? 0 : kStreamNever;
    }
    __syncthreads();   // the only workgroup barrier: nobody has left yet
    if (wave < kStreamWorkers && !worker) return;

    if (wave == kStreamWorkers) {
        // ---- helper: records of earlier workgroups global -> ring, then this workgroup's records ring -> global ----
        int fetched = 0, spins = 0;
#ifdef DMSA_STREAM_TIMING
        const long long th0 = wall_clock64();
        long long t_poll = 0;
        int rounds_f = 0, rounds_p = 0;
#endif
        while (fetched < s0) {
            unsigned long long seen;
#ifdef DMSA_STREAM_TIMING
            const long long tp = wall_clock64();
            ++rounds_f;
#endif
            if (lane == 0) {
                do {
                    seen = __hip_atomic_load(gw.published, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((seen >> 32) == (unsigned long long)epoch && (int)(seen & 0xffffffffull) > fetched) break;
                    stream_spin(spins);
                } while (true);
            }
            int avail = __builtin_amdgcn_readfirstlane(lane == 0 ? (int)(seen & 0xffffffffull) : 0);
            avail = min(avail, s0);
#ifdef DMSA_STREAM_TIMING
            t_poll += wall_clock64() - tp;
#endif
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the records below were stored before `published` moved (no write-back needed here)
            while (fetched < avail) {
                const int m = min(kStreamBatch, avail - fetched);
                double v[kStreamBatch][R];
                int pp[kStreamBatch];
#pragma unroll
                for (int u = 0; u < kStreamBatch; ++u) {
                    if (u < m) {
#pragma unroll
                        for (int i = 0; i < R; ++i) v[u][i] = gw.mult[(size_t)(fetched + u) * RN + lane + 64 * i];
                        pp[u] = gw.gpp[fetched + u];
                    }
                }
                // room in the ring: every worker is done with the records these overwrite
                while (stream_min_consumed(sm.consumed, lane) < fetched + m - kStreamD) stream_spin(spins);
#pragma unroll
                for (int u = 0; u < kStreamBatch; ++u) {
                    if (u < m) {
                        const int slot = (fetched + u) % kStreamD;
#pragma unroll
                        for (int i = 0; i < R; ++i) sm.ring_f[slot][lane + 64 * i] = v[u][i];
                        if (lane == 0) sm.ring_pp[slot] = pp[u];
                    }
                }
                fetched += m;
                __builtin_amdgcn_wave_barrier();
                lds_store_after_writes(&sm.produced, fetched);   // (every lane stores the same value)
            }
        }
        int copied = s0;
#ifdef DMSA_STREAM_TIMING
        const long long th1 = wall_clock64();
#endif
        while (copied < s1) {
#ifdef DMSA_STREAM_TIMING
            ++rounds_p;
#endif
            int have;
            while ((have = lds_load_acquire(&sm.produced)) <= copied) stream_spin(spins);
            have = __builtin_amdgcn_readfirstlane(have);
            for (int k = copied; k < have; ++k) {
                const int slot = k % kStreamD;
#pragma unroll
                for (int i = 0; i < R; ++i) gw.mult[(size_t)k * RN + lane + 64 * i] = sm.ring_f[slot][lane + 64 * i];
                if (lane == 0) gw.gpp[k] = sm.ring_pp[slot];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // once: the records are written back before `published` moves (it also waits for the ring reads; nothing to invalidate)
            if (lane == 0) {
                __hip_atomic_store(gw.published, ((unsigned long long)epoch << 32) | (unsigned long long)have, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&sm.consumed[kStreamWorkers], have, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            copied = have;
        }
#ifdef DMSA_STREAM_TIMING
        if (lane == 0)
            printf("[stream timing] helper of block %d: start %lld, %d records fetched in %d rounds, %lld (polling %lld); %d published in %d rounds, %lld, end %lld (x10 ns)\n", b,
                   th0, s0, rounds_f, th1 - th0, t_poll, s1 - s0, rounds_p, wall_clock64() - th1, wall_clock64());
#endif
        return;
    }

    // ---- worker ----
    double x[R][W];
    int logical[R];
    unsigned pivoted = 0;
    // All loads of the worker's columns first, then the arithmetic on them: with the masking in between, the compiler waited for every
    // column before it asked for the next one -- eight memory latencies in a row, 8 us before the first pivot step.  (The index is
    // clamped instead of the load being skipped, so there is no branch around a load either.)
    if (a_part) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = min(lane + 64 * i, n - 1);
#pragma unroll
            for (int c = 0; c < W; ++c) x[i][c] = Hp[(size_t)min(c0 + c, n - 1) * n1 + r];   // H(r, col)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int r = lane + 64 * i;
        logical[i] = r < n ? r : -1;
        if (r >= n) pivoted |= 1u << i;   // padding rows: never candidates, all zero
#pragma unroll
        for (int c = 0; c < W; ++c) {
            const int col = c0 + c;
            const bool in = r < n && col < n;
            if (a_part)
                x[i][c] = in ? (col == r ? x[i][c] + lambda : x[i][c]) : 0.0;   // damped on the diagonal (:110)
            else
                x[i][c] = in && col == r ? 1.0 : 0.0;
        }
    }
    double* scr = sm.scr[wave];
    // ---- apply the records of the steps before this worker's own columns (all of them for a worker of I columns) ----
    const int k_end = a_part ? c0 : n;
    int seen = 0, spins = 0;
#ifdef DMSA_STREAM_TIMING
    const long long t_begin = wall_clock64();
    long long t_wait = 0;
#endif
    for (int k = 0; k < k_end; ++k) {
        if (k >= seen) {
#ifdef DMSA_STREAM_TIMING
            const long long tw = wall_clock64();
#endif
            while ((seen = lds_load_acquire(&sm.produced)) <= k) stream_spin(spins);
            seen = __builtin_amdgcn_readfirstlane(seen);
#ifdef DMSA_STREAM_TIMING
            t_wait += wall_clock64() - tw;
#endif
        }
        const int slot = k % kStreamD;
        const int pp = __builtin_amdgcn_readfirstlane(sm.ring_pp[slot]);
        double f[R];
#pragma unroll
        for (int i = 0; i < R; ++i) f[i] = sm.ring_f[slot][lane + 64 * i];
        const double d = sm.ring_f[slot][pp];
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) lds_store_release(&sm.consumed[wave], k + 1);   // (release waits for the reads above)
        const int pl = pp & 63;
        const SlotFlags<R> sel = slot_flags<R>(pp >> 6);
        stream_swap_rows<R>(logical, pivoted, k, pl, sel, lane);
        stream_apply<R, 0>(x, f, d, pl, sel, lane, scr);
    }
#ifdef DMSA_STREAM_TIMING
    const long long t_consumed = wall_clock64();
    if (lane == 0 && !a_part && (wi == 0 || wi == 11))
        printf("[stream timing] I worker %d: start %lld end %lld, %d records in %lld (x10 ns), of which waiting %lld, spins %d\n", wi, t_begin, t_consumed, k_end,
               t_consumed - t_begin, t_wait, spins);
#endif
    if (!a_part) {
        // ---- columns of the inverse, transposed; the row permutation once ----
#pragma unroll
        for (int c = 0; c < W; ++c) {
            if (c0 + c < n) {
#pragma unroll
                for (int i = 0; i < R; ++i) gw.inv_t[(size_t)(c0 + c) * RN + lane + 64 * i] = x[i][c];
            }
        }
        if (wi == 0) {
#pragma unroll
            for (int i = 0; i < R; ++i) gw.iperm[lane + 64 * i] = logical[i];
        }
        return;
    }
    // ---- this worker's columns are the pivot columns now: it makes the records ----
    if (lane == 0) lds_store_release(&sm.consumed[wave], kStreamNever);   // it reads the ring no more
#define DMSA_STREAM_STEP(J) \
    if (c0 + J < n) stream_factor_step<R, J>(x, logical, pivoted, c0 + J, n, lane, sm, scr);
    DMSA_STREAM_STEP(0) DMSA_STREAM_STEP(1) DMSA_STREAM_STEP(2) DMSA_STREAM_STEP(3) DMSA_STREAM_STEP(4) DMSA_STREAM_STEP(5) DMSA_STREAM_STEP(6) DMSA_STREAM_STEP(7)
#if DMSA_STREAM_W > 8
    DMSA_STREAM_STEP(8) DMSA_STREAM_STEP(9) DMSA_STREAM_STEP(10) DMSA_STREAM_STEP(11) DMSA_STREAM_STEP(12) DMSA_STREAM_STEP(13) DMSA_STREAM_STEP(14) DMSA_STREAM_STEP(15)
#endif
#undef DMSA_STREAM_STEP
#ifdef DMSA_STREAM_TIMING
    if (lane == 0 && c0 + kStreamW >= n)
        printf("[stream timing] factor steps so far (clock64 ticks, all workers): search %lld  record %lld  bookkeeping %lld  apply %lld\n", g_stream_phase[0], g_stream_phase[1],
               g_stream_phase[2], g_stream_phase[3]);
    if (lane == 0)
        printf("[stream timing] A worker %d (block %d): start %lld, %d records applied in %lld (waiting %lld, spins %d), own columns factored in %lld (x10 ns)\n", wi, b,
               t_begin, k_end, t_consumed - t_begin, t_wait, spins, wall_clock64() - t_consumed);
#endif
}
// :113-128 behind k_loop_lm_stream: step = (-alpha H^-1) g row by row in column order (thread = physical row; the transposed inverse makes
// every load a coalesced one), logical order through iperm, NaN test and max_step clamp.
__global__ __launch_bounds__(256) void k_loop_lm_stream_tail(const double* __restrict__ Hp, int P, int RN, double alpha, double max_step, const double* __restrict__ work,
                                                             double* __restrict__ step, LoopFlags* __restrict__ flags) {
    if (flags->stop != 0) return;
    const int n = P, n1 = P + 1, r = threadIdx.x, lane = r & 63, wave = r >> 6;
    const StreamWork gw = stream_work(const_cast<double*>(work), P, RN);
    __shared__ double s_step[256];
    if (r < n) {
        const double* gv = Hp + (size_t)P * n1;
        double sres = 0.0;
        int j = 0;
        for (; j + 32 <= n; j += 32) {   // thirty-two loads in flight; the sum itself is a chain in column order
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = gw.inv_t[(size_t)(j + u) * RN + r];
#pragma unroll
            for (int u = 0; u < 32; ++u) sres += (-alpha * v[u]) * gv[j + u];
        }
        for (; j + 8 <= n; j += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = gw.inv_t[(size_t)(j + u) * RN + r];
#pragma unroll
            for (int u = 0; u < 8; ++u) sres += (-alpha * v[u]) * gv[j + u];
        }
        for (; j < n; ++j) sres += (-alpha * gw.inv_t[(size_t)j * RN + r]) * gv[j];
        s_step[gw.iperm[r]] = sres;
    }
    __syncthreads();
    if (wave == 0) {
        bool any_nan = false;
        double mx = -INFINITY, mn = INFINITY;
        for (int i = lane; i < n; i += 64) {
            const double v = s_step[i];
            any_nan = any_nan || isnan(v);
            mx = mx < v ? v : mx;
            mn = v < mn ? v : mn;
        }
        if (__ballot(any_nan) != 0ull) {
            if (lane == 0) flags->nan = 1;
        } else {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double a = __shfl_xor(mx, o), c = __shfl_xor(mn, o);
                mx = mx < a ? a : mx;
                mn = c < mn ? c : mn;
            }
            const double neg = -mn;
            const double max_elem = mx < neg ? neg : mx;
            for (int i = lane; i < n; i += 64) {
                const double v = s_step[i];
                step[i] = max_elem > max_step ? (max_step / max_elem) * v : v;
            }
        }
    }
}

// the same tail for a step the host solved (P > 64): one wave, NaN test and extrema as wave reductions (max / min are order independent)
__global__ __launch_bounds__(kWave) void k_loop_step_finish(int P, double max_step, double* __restrict__ step, LoopFlags* __restrict__ flags) {
    if (flags->stop != 0) return;
    const int lane = threadIdx.x;
    bool any_nan = false;
    double mx = -INFINITY, mn = INFINITY;
    for (int i = lane; i < P; i += kWave) {
        const double v = step[i];
        any_nan = any_nan || isnan(v);
        mx = mx < v ? v : mx;
        mn = v < mn ? v : mn;
    }
    if (__ballot(any_nan) != 0ull) {
        if (lane == 0) flags->nan = 1;
        return;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double a = __shfl_xor(mx, o), b = __shfl_xor(mn, o);
        mx = mx < a ? a : mx;
        mn = b < mn ? b : mn;
    }
    const double neg = -mn;
    const double max_elem = mx < neg ? neg : mx;  // std::max(mx, -mn)
    if (max_elem > max_step)
        for (int i = lane; i < P; i += kWave) step[i] = (max_step / max_elem) * step[i];
}

__device__ __forceinline__ void loop_finish_body(double* sm, const LoopModel& m, const double* __restrict__ state_jac, const double* __restrict__ state_trial,
                                                 double* __restrict__ state0, double* __restrict__ paramVec, const double* __restrict__ step,
                                                 const double* __restrict__ error0_ptr, const double* __restrict__ errs, int errs_nsplit, int fixed_iters,
                                                 double epsilon, IterResult* __restrict__ result, LoopFlags* __restrict__ flags, double* __restrict__ ctrl0,
                                                 int chain_next) {
    if (flags->stop != 0) return;
    const int n = m.n, P = m.P;
    const ChainLds c = carve(sm, n, P);
    __shared__ int s_best, s_stop;
    __shared__ double s_norm;
    const double error0 = *error0_ptr;  // e0^T e0 (:101)
    if (flags->nan != 0) {
        // :116-122 setPoseParameters(paramVec); break -- on the state the Jacobian batch left
        load_state(state_jac, n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
        set_params_lds(n, paramVec, c.rel_o, c.rel_t);
        if (m.model == 2) wave_relative_to_global(n, c.rel_o, c.rel_t, c.glob_o, c.glob_t, c.E, c.R);
        store_state(state0, n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
        if (threadIdx.x == 0) {
            result->stop = 2 /* DMSA_STOP_NAN */, result->best_k = 0, result->error0 = error0, result->step_norm = 0.0;
            flags->stop = 2;
        }
        return;
    }
    // the step and the nine trial errors through LDS (coalesced loads; the sums below are serial chains on one lane)
    for (int i = threadIdx.x; i < P; i += kWave) c.org[i] = step[i];
    if (threadIdx.x < 9) {
        // e^T e of trial threadIdx.x + 1: reduced already (errs_nsplit == 0), or its block sums added here in block order
        double e = 0.0;
        if (errs_nsplit <= 0) {
            e = errs[threadIdx.x];
        } else {
            const double* src = errs + (size_t)threadIdx.x * errs_nsplit;
            int sp = 0;
            for (; sp + 16 <= errs_nsplit; sp += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[sp + u];
#pragma unroll
                for (int u = 0; u < 16; ++u) e += v[u];
            }
            for (; sp < errs_nsplit; ++sp) e += src[sp];
        }
        c.E[threadIdx.x] = e;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double min_error = error0;
        int best = 0;
        for (int k = 1; k < 10; ++k)
            if (c.E[k - 1] < min_error) min_error = c.E[k - 1], best = k;
        double ss = 0.0;
        for (int i = 0; i < P; ++i) ss += c.org[i] * c.org[i];
        const double norm = sqrt(ss);
        int stop = 0;
        if (best == 0 && !fixed_iters)
            stop = 3;  // DMSA_STOP_NO_IMPROVEMENT: the set stays at raw + 0.9 step, not restored (:130-134)
        else if (norm < epsilon && !fixed_iters)
            stop = 4;  // DMSA_STOP_EPSILON (:139-143), after setPoseParameters
        s_best = best, s_stop = stop, s_norm = norm;
        result->stop = stop, result->best_k = best, result->error0 = error0, result->step_norm = norm;
        if (stop != 0) flags->stop = stop;
    }
    __syncthreads();
    load_state(state_trial, n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
    if (s_stop != 3) {
        // :136 setPoseParameters(best parameters): raw + 0.1 k step for the winning k, raw when no trial won (fixed iterations only)
        const int best = s_best;
        for (int i = threadIdx.x; i < P; i += kWave) c.par[i] = best > 0 ? paramVec[i] + 0.1 * (double)best * c.org[i] : paramVec[i];
        __syncthreads();
        set_params_lds(n, c.par, c.rel_o, c.rel_t);
        if (m.model == 2) wave_relative_to_global(n, c.rel_o, c.rel_t, c.glob_o, c.glob_t, c.E, c.R);
    }
    store_state(state0, n, c.rel_o, c.rel_t, c.glob_o, c.glob_t);
    // the loop goes on: the start of the next iteration (what k_loop_begin does for the first) right here, one launch less on the
    // critical path.  Not after the last iteration: decentralize() reads the global poses of the last evaluated trial.
    if (s_stop == 0 && chain_next) {
        __syncthreads();
        begin_body(m, c, state0, paramVec, ctrl0);
    }
}
__global__ __launch_bounds__(kWave) void k_loop_finish(const LoopModel m, const double* __restrict__ state_jac, const double* __restrict__ state_trial,
                                                       double* __restrict__ state0, double* __restrict__ paramVec, const double* __restrict__ step,
                                                       const double* __restrict__ error0_ptr, const double* __restrict__ errs, int errs_nsplit,
                                                       int fixed_iters, double epsilon, IterResult* __restrict__ result, LoopFlags* __restrict__ flags,
                                                       double* __restrict__ ctrl0, int chain_next, uint32_t* state_ready) {
    extern __shared__ double sm[];
    loop_finish_body(sm, m, state_jac, state_trial, state0, paramVec, step, error0_ptr, errs, errs_nsplit, fixed_iters, epsilon, result, flags, ctrl0, chain_next);
    DevSync sy;
    sy.signal_counter = state_ready;  // the next iteration's Jacobian chains (side stream) may start: on every path, a stopped loop included
    dev_sync_leave(sy);
}

}  // namespace

// The chain kernels keep one chain state (30 n + 2 P + rows doubles) in dynamic LDS.  Up to 64 KB that needs nothing; beyond, the limit of
// the three kernels is raised once (gfx950: 160 KB per workgroup), and a set that does not fit even then (more than ~470 keyframes) is
// left to the host-driven loop by optimize() -- loop_chain_fits() is what it asks.
constexpr size_t kChainLdsMax = 160 * 1024 - 256;
bool loop_chain_fits(const LoopModel& m) { return chain_lds_bytes(m) <= kChainLdsMax; }
static void chain_lds_allow(size_t bytes) {
    if (bytes <= 64 * 1024) return;  // (per launch: the attribute belongs to the current device, and sets this large are rare)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop_begin), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLdsMax);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop_chain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLdsMax);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop_finish), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainLdsMax);
}
void launch_loop_begin(const LoopModel& m, double* state0, double* paramVec, double* ctrl0, LoopFlags* flags, hipStream_t s, uint32_t* state_ready) {
    chain_lds_allow(chain_lds_bytes(m));
    hipLaunchKernelGGL(k_loop_begin, dim3(1), dim3(kWave), chain_lds_bytes(m), s, m, state0, paramVec, ctrl0, flags, state_ready);
}
void launch_loop_chain(const LoopModel& m, int mode, const double* state_in, double* state_out, const double* paramVec, const double* step, double increment,
                       double* ctrl, double* extra, const LoopFlags* flags, hipStream_t s, int part, uint32_t* start_signal) {
    const int B = mode == 0 ? 1 + m.P : 9;
    chain_lds_allow(chain_lds_bytes(m));
    hipLaunchKernelGGL(k_loop_chain, dim3(B), dim3(kWave), chain_lds_bytes(m), s, m, mode, state_in, state_out, paramVec, step, increment, ctrl, extra, flags,
                       mode == 1 ? part : 0, start_signal);
}
void launch_loop_scatter_extra(const double* extra, int B, int a, double* E, int64_t ldE, int M, hipStream_t s) {
    if (a <= 0 || B <= 0) return;
    hipLaunchKernelGGL(k_loop_scatter_extra, dim3((B * a + 255) / 256), dim3(256), 0, s, extra, B, a, E, ldE, M);
}
static void launch_lm_step(const HpSource& hp, int P, double lambda, double alpha, double max_step, double* step, LoopFlags* flags, double* error0_out,
                           hipStream_t s) {
    const size_t bytes = (2 * (size_t)P * P + (size_t)P) * sizeof(double);
    static bool raised = false;
    if (bytes > 48 * 1024 && !raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_loop_lm_step<2, kSolveRows>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        raised = true;
    }
    if (P <= 4 * kSolveWaves)  // the default window (P = 30): one column per lane, four rows per wave
        hipLaunchKernelGGL((k_loop_lm_step<1, 4>), dim3(1), dim3(kSolveWaves * kWave), bytes, s, hp, P, lambda, alpha, max_step, step, flags, error0_out);
    else
        hipLaunchKernelGGL((k_loop_lm_step<2, kSolveRows>), dim3(1), dim3(kSolveWaves * kWave), bytes, s, hp, P, lambda, alpha, max_step, step, flags, error0_out);
}
void launch_loop_lm_step(const double* Hp, int P, double lambda, double alpha, double max_step, double* step, LoopFlags* flags, hipStream_t s) {
    launch_lm_step(HpSource{Hp, nullptr, 0, 0}, P, lambda, alpha, max_step, step, flags, nullptr, s);
}
void launch_loop_lm_step_partials(const double* partial, int nsplit, int nt, int P, double lambda, double alpha, double max_step, double* step, LoopFlags* flags,
                                  double* error0_out, hipStream_t s) {
    launch_lm_step(HpSource{nullptr, partial, nsplit, nt}, P, lambda, alpha, max_step, step, flags, error0_out, s);
}
static int stream_rows_per_lane(int P) { return (P + 63) / 64; }
size_t loop_panel_solve_doubles(int P) {
    const size_t panels = ((size_t)P + kPanel - 1) / kPanel;
    const size_t blocked = panels * (size_t)P * kPanel /* published panels */ + (size_t)P * P /* inverse */ + panels + 2 /* flags, done counter (as 8-byte words) */ +
                           (size_t)P / 2 + 2 /* final row permutation (ints) */;
    const size_t RN = 64 * (size_t)stream_rows_per_lane(P);
    const size_t stream = 2 * (size_t)P * RN /* records, transposed inverse */ + 2 /* published */ + (size_t)P / 2 + 2 /* pivot rows (ints) */ + RN / 2 + 2 /* permutation */;
    return blocked > stream ? blocked : stream;
}
bool loop_lm_stream_fits(int P) { return P > kLoopSolveMaxP && P <= kLoopStreamMaxP; }
void launch_loop_lm_stream(const double* Hp, int P, double lambda, double alpha, double max_step, double* work, unsigned int epoch, double* step,
                           LoopFlags* flags, hipStream_t s) {
    const int R = stream_rows_per_lane(P), RN = 64 * R;
    const int nA = (P + kStreamW - 1) / kStreamW, GA = (nA + kStreamWorkers - 1) / kStreamWorkers;
    const dim3 grid(2 * GA), block((kStreamWorkers + 1) * kWave);
    if (R <= 2)
        hipLaunchKernelGGL(k_loop_lm_stream<2>, grid, block, 0, s, Hp, P, lambda, work, epoch, flags);
    else
        hipLaunchKernelGGL(k_loop_lm_stream<3>, grid, block, 0, s, Hp, P, lambda, work, epoch, flags);
    hipLaunchKernelGGL(k_loop_lm_stream_tail, dim3(1), dim3(RN), 0, s, Hp, P, RN, alpha, max_step, work, step, flags);
}
void launch_loop_lm_panels(const double* Hp, int P, double lambda, double alpha, double max_step, double* work, unsigned int epoch, double* step,
                           LoopFlags* flags, hipStream_t s) {
    const int nblocks = (2 * P + kPanel - 1) / kPanel;
    const int threads = ((P + 63) / 64) * 64;
    const size_t lds = kPanelLdsHead * sizeof(double) + ((size_t)P + 2) / 2 * sizeof(double) + sizeof(double) + (size_t)P * sizeof(double) + 64;
    hipLaunchKernelGGL(k_loop_lm_panels, dim3(nblocks), dim3(threads), lds, s, Hp, P, lambda, alpha, max_step, work, epoch, step, flags);
}
// ---- stream dependencies without barrier packets (dev_sync.h) ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_sync_signal(uint32_t* counter) {
    if (threadIdx.x == 0) dev_sync_signal(counter);
}
__global__ __launch_bounds__(64) void k_sync_wait(const uint32_t* counter, uint32_t target, int32_t* timed_out, int max_spins) {
    if (threadIdx.x == 0) dev_sync_wait(counter, target, timed_out, max_spins);
}
// debug switch gap_stamps: the device's 100 MHz wall clock at this point of a stream (a one-thread kernel between the kernels in question)
__global__ __launch_bounds__(64) void k_stamp(long long* slot) {
    if (threadIdx.x == 0) *slot = wall_clock64();
}
void launch_stamp(long long* slot, hipStream_t s) { hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, slot); }
void launch_sync_signal(uint32_t* counter, hipStream_t s) { hipLaunchKernelGGL(k_sync_signal, dim3(1), dim3(64), 0, s, counter); }
void launch_sync_wait(const uint32_t* counter, uint32_t target, int32_t* timed_out, hipStream_t s, int max_spins) {
    hipLaunchKernelGGL(k_sync_wait, dim3(1), dim3(64), 0, s, counter, target, timed_out, max_spins);
}
void launch_loop_step_finish(int P, double max_step, double* step, LoopFlags* flags, hipStream_t s) {
    hipLaunchKernelGGL(k_loop_step_finish, dim3(1), dim3(64), 0, s, P, max_step, step, flags);
}
void launch_loop_finish(const LoopModel& m, const double* state_jac, const double* state_trial, double* state0, double* paramVec, const double* step,
                        const double* error0, const double* trial_errs, int trial_nsplit, int fixed_iters, double epsilon, IterResult* result, LoopFlags* flags,
                        double* ctrl0, int chain_next, hipStream_t s, uint32_t* state_ready) {
    chain_lds_allow(chain_lds_bytes(m));
    hipLaunchKernelGGL(k_loop_finish, dim3(1), dim3(kWave), chain_lds_bytes(m), s, m, state_jac, state_trial, state0, paramVec, step, error0, trial_errs,
                       trial_nsplit, fixed_iters, epsilon, result, flags, ctrl0, chain_next, state_ready);
}

}  // namespace dmsa
