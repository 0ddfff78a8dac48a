// host_math.cpp — see host_math.h.  Compiled with -ffp-contract=off so that the double sequences here are
// reproducible across compilers (the pose table built here feeds float casts and voxel keys downstream).
#include "host_math.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>

#include "../../include/dmsa_detmath.h"

#include <algorithm>
#include <cstring>
#include <limits>

namespace dmsa {

namespace {
struct Quat {
    double w, x, y, z;
};
inline Quat quat_from_axang(Vec3 a) {
    const double sq = a.x * a.x + a.y * a.y + a.z * a.z;
    const double ang = std::sqrt(sq);
    Vec3 ax = a;
    if (sq > 0.0) ax = {a.x / ang, a.y / ang, a.z / ang};
    const double sh = dmsa_det::det_sin(0.5 * ang);
    return {dmsa_det::det_cos(0.5 * ang), sh * ax.x, sh * ax.y, sh * ax.z};
}
}  // namespace

Vec3 slerp_axang(Vec3 a, Vec3 b, double t) {
    const Quat q1 = quat_from_axang(a), q2 = quat_from_axang(b);
    const double one = 1.0 - std::numeric_limits<double>::epsilon();
    const double d = q1.w * q2.w + q1.x * q2.x + q1.y * q2.y + q1.z * q2.z;
    const double ad = std::fabs(d);
    double s0, s1;
    if (ad >= one) {
        s0 = 1.0 - t, s1 = t;
    } else {
        const double th = dmsa_det::det_acos(ad), sn = dmsa_det::det_sin(th);
        s0 = dmsa_det::det_sin((1.0 - t) * th) / sn;
        s1 = dmsa_det::det_sin(t * th) / sn;
    }
    if (d < 0.0) s1 = -s1;
    const Quat q{s0 * q1.w + s1 * q2.w, s0 * q1.x + s1 * q2.x, s0 * q1.y + s1 * q2.y, s0 * q1.z + s1 * q2.z};
    double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    if (n == 0.0) return {0.0, 0.0, 0.0};
    const double angle = 2.0 * dmsa_det::det_atan2(n, std::fabs(q.w));
    if (q.w < 0.0) n = -n;
    return {(q.x / n) * angle, (q.y / n) * angle, (q.z / n) * angle};
}

// ---- PoseChain ------------------------------------------------------------------------------------
void PoseChain::resize(int count) {
    n = count;
    rel_o.assign(3 * (size_t)n, 0.0), rel_t.assign(3 * (size_t)n, 0.0);
    glob_o.assign(3 * (size_t)n, 0.0), glob_t.assign(3 * (size_t)n, 0.0);
}
static inline Vec3 col(const std::vector<double>& m, int k) { return {m[3 * k], m[3 * k + 1], m[3 * k + 2]}; }

void PoseChain::relative_to_global() { chain_relative_to_global(n, rel_o.data(), rel_t.data(), glob_o.data(), glob_t.data()); }
void PoseChain::global_to_relative() { chain_global_to_relative(n, glob_o.data(), glob_t.data(), rel_o.data(), rel_t.data()); }
void PoseChain::get_params(double* p) const {
    std::copy(rel_o.begin() + 3, rel_o.end(), p);
    std::copy(rel_t.begin() + 3, rel_t.end(), p + 3 * (n - 1));
}
void PoseChain::set_params(const double* p) {
    std::copy(p, p + 3 * (n - 1), rel_o.begin() + 3);
    std::copy(p + 3 * (n - 1), p + 6 * (n - 1), rel_t.begin() + 3);
}

// ---- FloaterHormann2 --------------------------------------------------------------------------------
bool FloaterHormann2::build(const double* nodes, int n) {
    const int d = 2;
    x.assign(nodes, nodes + n);
    w.assign((size_t)n, 0.0);
    for (int k = 0; k < n; ++k) {
        const int i_lo = std::max(k - d, 0);
        const int i_hi = (k >= n - d) ? n - d - 1 : k;
        for (int i = i_lo; i <= i_hi; ++i) {
            double prod = 1.0;
            const int j_hi = std::min(i + d, n - 1);
            for (int j = i; j <= j_hi; ++j) {
                if (j == k) continue;
                const double diff = x[k] - x[j];
                if (std::fabs(diff) < std::numeric_limits<double>::min()) return false;
                prod *= diff;
            }
            if (i % 2 == 0)
                w[k] += 1.0 / prod;
            else
                w[k] -= 1.0 / prod;
        }
    }
    return true;
}
double FloaterHormann2::eval(const double* y, double t) const {
    double num = 0.0, den = 0.0;
    for (size_t i = 0; i < x.size(); ++i) {
        if (t == x[i]) return y[i];
        const double q = w[i] / (t - x[i]);
        num += q * y[i];
        den += q;
    }
    return num / den;
}

// ---- dense tables -----------------------------------------------------------------------------------
static inline void store_row(float* T, const Mat3& R, Vec3 t) {
    T[0] = (float)R(0, 0), T[1] = (float)R(0, 1), T[2] = (float)R(0, 2), T[3] = (float)t.x;
    T[4] = (float)R(1, 0), T[5] = (float)R(1, 1), T[6] = (float)R(1, 2), T[7] = (float)t.y;
    T[8] = (float)R(2, 0), T[9] = (float)R(2, 1), T[10] = (float)R(2, 2), T[11] = (float)t.z;
}

void window_dense_table(const PoseChain& ctrl, const std::vector<double>& stamps, const FloaterHormann2& fh,
                        const std::vector<double>& traj_time, float* table) {
    const int C = ctrl.n;
    const int n_t = (int)traj_time.size();
    std::vector<double> ax((size_t)C), ay((size_t)C), az((size_t)C);
    for (int k = 0; k < C; ++k) ax[k] = ctrl.glob_t[3 * k], ay[k] = ctrl.glob_t[3 * k + 1], az[k] = ctrl.glob_t[3 * k + 2];
    for (int j = 0; j < n_t; ++j) {
        const double t = traj_time[j];
        // getInterpRotation (ContinuousTrajectory.h:570-591): lower_bound over all stamps but the last
        const int right = (int)(std::lower_bound(stamps.begin(), stamps.end() - 1, t) - stamps.begin());
        Vec3 o;
        if (right > 0) {
            const double t_rel = (t - stamps[right - 1]) / (stamps[right] - stamps[right - 1]);
            o = slerp_axang(col(ctrl.glob_o, right - 1), col(ctrl.glob_o, right), t_rel);
        } else {
            o = col(ctrl.glob_o, 0);
        }
        const Vec3 tr{fh.eval(ax.data(), t), fh.eval(ay.data(), t), fh.eval(az.data(), t)};
        store_row(table + 12 * (size_t)j, so3_exp(o), tr);
    }
}

void keyframe_table(const PoseChain& frames, float* table) {
    for (int k = 0; k < frames.n; ++k) store_row(table + 12 * (size_t)k, so3_exp(col(frames.glob_o, k)), col(frames.glob_t, k));
}

// ---- WindowHost ---------------------------------------------------------------------------------------
bool WindowHost::init(const dmsa_window_problem& p) {
    const int C = p.num_control_poses;
    // Floater-Hormann with d = 2 needs at least three nodes (boost's barycentric_rational throws for n <= order; with two nodes every
    // weight would be zero and the tables NaN)
    if (C < 3 || p.n_total < 2) return false;
    if (!p.rel_orient || !p.rel_transl || !p.stamps || !p.traj_time) return false;
    ctrl.resize(C);
    std::copy(p.rel_orient, p.rel_orient + 3 * C, ctrl.rel_o.begin());
    std::copy(p.rel_transl, p.rel_transl + 3 * C, ctrl.rel_t.begin());
    stamps.assign(p.stamps, p.stamps + C);
    traj_time.assign(p.traj_time, p.traj_time + p.n_total);
    if (!fh.build(stamps.data(), C)) return false;
    use_imu = p.use_imu != 0;
    if (use_imu) {
        if (!p.param_indices || !p.preint_rot || !p.preint_pos || !p.preint_vel || !p.cov_pvrot_inv || !(p.dt_res > 0.0)) return false;
        // updateImuError reads trajTime[i0 + 1] and trajTime[i1 - 1] (ContinuousTrajectory.h:634-637): the indices must stay inside the grid
        for (int k = 1; k < C; ++k) {
            const int i0 = p.param_indices[k - 1], i1 = p.param_indices[k];
            if (i0 < 0 || i0 + 1 >= p.n_total || i1 < 1 || i1 >= p.n_total) return false;
        }
        dt_res = p.dt_res, balancing_imu = p.balancing_imu;
        gravity = {p.gravity[0], p.gravity[1], p.gravity[2]};
        param_indices.assign(p.param_indices, p.param_indices + C);
        preint_rot.assign(p.preint_rot, p.preint_rot + 9 * C);
        preint_pos.assign(p.preint_pos, p.preint_pos + 3 * C);
        preint_vel.assign(p.preint_vel, p.preint_vel + 3 * C);
        cov_inv.assign(p.cov_pvrot_inv, p.cov_pvrot_inv + 81 * C);
    }
    return true;
}

ImuConsts WindowHost::imu_consts() const {
    ImuConsts c{};
    c.use_imu = use_imu ? 1 : 0, c.dt_res = dt_res, c.balancing_imu = balancing_imu;
    c.gravity[0] = gravity.x, c.gravity[1] = gravity.y, c.gravity[2] = gravity.z;
    c.param_indices = param_indices.data(), c.preint_rot = preint_rot.data(), c.preint_pos = preint_pos.data(), c.preint_vel = preint_vel.data();
    c.cov_inv = cov_inv.data();
    return c;
}
void WindowHost::imu_rows(double* rows) {
    ctrl.global_to_relative();  // ContinuousTrajectory.h:606
    const ImuConsts c = imu_consts();
    for (int k = 1; k < ctrl.n; ++k)
        rows[k - 1] = imu_row(k, ctrl.n, stamps.data(), fh.w.data(), traj_time.data(), c, ctrl.glob_o.data(), ctrl.glob_t.data(), col(ctrl.rel_o, k));
}

// ---- KeyframeHost -------------------------------------------------------------------------------------
bool KeyframeHost::init(const dmsa_keyframe_problem& p) {
    const int F = p.num_frames;
    if (F < 2) return false;
    frames.resize(F);
    std::copy(p.rel_orient, p.rel_orient + 3 * F, frames.rel_o.begin());
    std::copy(p.rel_transl, p.rel_transl + 3 * F, frames.rel_t.begin());
    frames.relative_to_global();
    use_gravity = p.use_gravity != 0, use_odometry = p.use_odometry != 0;
    gravity = {p.gravity[0], p.gravity[1], p.gravity[2]};
    std::copy(p.cov_grav_inv, p.cov_grav_inv + 9, cov_grav_inv);
    balancing_grav = p.balancing_grav, balancing_odom = p.balancing_odom;
    if (use_gravity) {
        measured_gravity.assign(p.measured_gravity, p.measured_gravity + 3 * F);
        gravity_plausible.assign(p.gravity_plausible, p.gravity_plausible + F);
    }
    if (use_odometry) {
        odom_transl.assign(p.odom_rel_transl, p.odom_rel_transl + 3 * F);
        odom_orient_mat.assign(p.odom_rel_orient_mat, p.odom_rel_orient_mat + 9 * F);
        std::copy(p.odom_transl_cov_inv, p.odom_transl_cov_inv + 9, odom_transl_cov_inv);
        std::copy(p.odom_orient_cov_inv, p.odom_orient_cov_inv + 9, odom_orient_cov_inv);
    }
    return true;
}
int KeyframeHost::num_extra_rows() const { return (use_gravity ? frames.n : 0) + (use_odometry ? frames.n - 1 : 0); }

KeyframeRowConsts KeyframeHost::row_consts() const {
    KeyframeRowConsts c{};
    c.use_gravity = use_gravity ? 1 : 0, c.use_odometry = use_odometry ? 1 : 0;
    c.gravity[0] = gravity.x, c.gravity[1] = gravity.y, c.gravity[2] = gravity.z;
    std::copy(cov_grav_inv, cov_grav_inv + 9, c.cov_grav_inv);
    c.balancing_grav = balancing_grav, c.balancing_odom = balancing_odom;
    std::copy(odom_transl_cov_inv, odom_transl_cov_inv + 9, c.odom_transl_cov_inv);
    std::copy(odom_orient_cov_inv, odom_orient_cov_inv + 9, c.odom_orient_cov_inv);
    c.measured_gravity = measured_gravity.data(), c.gravity_plausible = gravity_plausible.data();
    c.odom_transl = odom_transl.data(), c.odom_orient_mat = odom_orient_mat.data();
    return c;
}
void KeyframeHost::additional_rows(double* rows) const {
    const int F = frames.n;
    const KeyframeRowConsts c = row_consts();
    int at = 0;
    if (use_gravity) {  // MapManagement.h:210-232 — row 0 and implausible frames stay exactly 0
        for (int k = 0; k < F; ++k) rows[at + k] = gravity_row(k, c, col(frames.glob_o, k));
        at += F;
    }
    if (use_odometry)  // MapManagement.h:234-252
        for (int k = 1; k < F; ++k) rows[at + k - 1] = odometry_row(k, c, col(frames.rel_o, k), col(frames.rel_t, k));
}

// ---- LM solve -----------------------------------------------------------------------------------------
namespace {
// One cache line per worker: a worker announces "I finished phase k" by storing k, and waits until everybody did.  No shared
// read-modify-write, so a barrier costs about one cache-to-cache transfer (a shared counter cost 5-8 us per barrier on the 2-socket
// EPYC host).  The workers of a solve meet once per pivot step, 1-2 us apart, so they spin (and yield if the machine is oversubscribed).
struct alignas(64) PhaseFlag {
    std::atomic<int> phase{0};
};
constexpr int kMaxSolveThreads = 16;
}  // namespace

namespace {
// x[i] -= f * s[i] and x[i] /= d: separate IEEE operations per element (this file is compiled with -ffp-contract=off), so the vector
// width does not change a bit; the AVX2 copies are only taken on CPUs that have them.
__attribute__((target("avx2"))) void row_sub_scaled_avx2(double* __restrict__ x, const double* __restrict__ s, double f, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] -= f * s[i];
}
void row_sub_scaled_base(double* __restrict__ x, const double* __restrict__ s, double f, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] -= f * s[i];
}
__attribute__((target("avx2"))) void row_div_avx2(double* __restrict__ x, double d, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] /= d;
}
void row_div_base(double* __restrict__ x, double d, size_t n) {
    for (size_t i = 0; i < n; ++i) x[i] /= d;
}
inline void row_sub_scaled(double* x, const double* s, double f, size_t n, bool avx2) { avx2 ? row_sub_scaled_avx2(x, s, f, n) : row_sub_scaled_base(x, s, f, n); }
inline void row_div(double* x, double d, size_t n, bool avx2) { avx2 ? row_div_avx2(x, d, n) : row_div_base(x, d, n); }
}  // namespace

void lm_solve(const double* Hin, const double* g, int P, double alpha, double* step, const ParallelRun* par, int max_threads) {
    // Gauss-Jordan with partial pivoting on [A | I]; the same operation on every element that reaches the result as the column-major
    // statement of the oracle (invert_dense), but stored row-major so that the row updates are contiguous.  Columns of A left of the
    // pivot hold exact zeros after their own elimination step (x - x*1) and are never read again, so they are not updated.
    const size_t n = (size_t)P;
    std::vector<double> A(n * n), inv(n * n, 0.0);
    bool finite = true;
    for (size_t r = 0; r < n; ++r)
        for (size_t c = 0; c < n; ++c) {
            A[r * n + c] = Hin[c * n + r];  // element (r, c) of the column-major input
            finite = finite && std::isfinite(Hin[c * n + r]);
        }
    for (size_t i = 0; i < n; ++i) inv[i * n + i] = 1.0;
    if (par == nullptr || P < 64 || !finite) {
        for (size_t c0 = 0; c0 < n; ++c0) {
            size_t piv = c0;
            double best = std::fabs(A[c0 * n + c0]);
            for (size_t r = c0 + 1; r < n; ++r)
                if (std::fabs(A[r * n + c0]) > best) best = std::fabs(A[r * n + c0]), piv = r;
            if (piv != c0) {
                std::swap_ranges(&A[c0 * n + c0], &A[c0 * n] + n, &A[piv * n + c0]);
                std::swap_ranges(&inv[c0 * n], &inv[c0 * n] + n, &inv[piv * n]);
            }
            double* a0 = &A[c0 * n];
            double* i0 = &inv[c0 * n];
            const double d = a0[c0];
            for (size_t c = c0; c < n; ++c) a0[c] /= d;
            for (size_t c = 0; c < n; ++c) i0[c] /= d;
            for (size_t r = 0; r < n; ++r) {
                if (r == c0) continue;
                double* ar = &A[r * n];
                double* ir = &inv[r * n];
                const double f = ar[c0];
                if (f == 0.0) continue;
                for (size_t c = c0; c < n; ++c) ar[c] -= f * a0[c];
                for (size_t c = 0; c < n; ++c) ir[c] -= f * i0[c];
            }
        }
        for (size_t i = 0; i < n; ++i) {
            double s = 0.0;
            const double* ii = &inv[i * n];
            for (size_t j = 0; j < n; ++j) s += (-alpha * ii[j]) * g[j];  // element (i, j) of the inverse
            step[i] = s;
        }
        return;
    }
    // P >= 64 (the keyframe pass, P = 186 per 32-frame neighbourhood, where the solve was 60 % of an iteration): blocked, on the
    // caller's worker threads, with the SAME operations on every element.  A pivot step k turns element x of a row into x - f_rk s_kc
    // (f_rk: the row's entry in column k before the step, s_kc: the scaled pivot row) or, in the pivot row, into x / d_k.  K = 8
    // consecutive steps are applied to an element in one go, in step order, so its sequence of roundings is unchanged; what the steps
    // need -- pivots, d_k, f_rk -- depends only on the K panel columns, which every worker factors privately (n x K numbers), as it
    // does the K scaled pivot rows.  Then every worker updates its own block of rows from the previous buffer into the next one
    // (two buffers: nobody overwrites what a slower worker still reads), one barrier per panel: 24 barriers instead of 186, and the
    // row updates stay in L1.  Row swaps are a permutation every worker tracks privately.
    constexpr int K = 8;
    const size_t ld = 2 * n;
    std::vector<double> buf[2] = {std::vector<double>(n * ld), std::vector<double>(n * ld)};
    for (size_t r = 0; r < n; ++r) {
        std::copy(&A[r * n], &A[r * n] + n, &buf[0][r * ld]);
        std::copy(&inv[r * n], &inv[r * n] + n, &buf[0][r * ld + n]);
    }
    const bool avx2 = __builtin_cpu_supports("avx2");
    PhaseFlag flags[kMaxSolveThreads];
    constexpr bool trace = false;  // flip for a phase timing of the blocked solve on stderr
    const auto t_enter = std::chrono::steady_clock::now();
    double tr_us[4] = {0, 0, 0, 0};
    (*par)([&](int t, int nthr) {
        const int use = std::min({nthr, std::max(1, std::min(kMaxSolveThreads, max_threads)), std::max(1, P / 16)});
        if (t >= use) return;
        int phase = 0;
        auto barrier = [&]() {
            ++phase;
            flags[t].phase.store(phase, std::memory_order_release);
            for (int u = 0; u < use; ++u)
                for (int spins = 0; flags[u].phase.load(std::memory_order_acquire) < phase; ++spins) {
                    if (spins > 4096)
                        std::this_thread::yield();
                    else
                        __builtin_ia32_pause();
                }
        };
        const size_t lo = n * (size_t)t / (size_t)use, hi = n * (size_t)(t + 1) / (size_t)use;
        std::vector<int> perm(n), iperm(n);  // logical row -> physical row and back
        std::vector<char> pivoted(n, 0);
        for (size_t i = 0; i < n; ++i) perm[i] = iperm[i] = (int)i;
        std::vector<double> Pn(n * K), F(n * K), S((size_t)K * ld);
        int piv[K];
        double dd[K];
        int panel = 0;
        for (size_t k0 = 0; k0 < n; k0 += K, ++panel) {
            const size_t kp = std::min((size_t)K, n - k0), k1 = k0 + kp;
            const double* src = buf[panel & 1].data();
            double* dst = buf[(panel + 1) & 1].data();
            barrier();  // the previous panel's output is complete
            if (trace && t == 0 && panel == 0) tr_us[0] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count();
            // ---- the panel columns of all rows, factored privately (every worker gets the same pivots and multipliers) ----
            for (size_t r = 0; r < n; ++r)
                for (size_t j = 0; j < (size_t)K; ++j) Pn[r * K + j] = j < kp ? src[r * ld + k0 + j] : 0.0;
            for (size_t j = 0; j < kp; ++j) {
                const size_t k = k0 + j;
                int pl = -1;
                double best = 0.0;
                for (size_t r = 0; r < n; ++r) {  // first strict maximum in logical row order = the serial search from row k down
                    if (pivoted[r]) continue;
                    const double v = std::fabs(Pn[r * K + j]);
                    const int l = iperm[r];
                    if (pl < 0 || v > best || (v == best && l < pl)) best = v, pl = l;
                }
                const int pp = perm[(size_t)pl];
                const int p0 = perm[k];
                perm[k] = pp, perm[(size_t)pl] = p0, iperm[(size_t)pp] = (int)k, iperm[(size_t)p0] = pl;
                pivoted[(size_t)pp] = 1;
                piv[j] = pp;
                const double d = Pn[(size_t)pp * K + j];
                dd[j] = d;
                // whole panel rows (fixed length K: vectorised): the columns left of j hold finished entries nobody reads again
                double prow[K];
                for (int c = 0; c < K; ++c) prow[c] = Pn[(size_t)pp * K + c] / d;
                for (int c = 0; c < K; ++c) Pn[(size_t)pp * K + c] = prow[c];
                for (size_t r = 0; r < n; ++r) {
                    if ((int)r == pp) continue;
                    double* pr = &Pn[r * K];
                    const double f = pr[j];
                    F[r * K + j] = f;
                    if (f == 0.0) continue;
                    for (int c = 0; c < K; ++c) pr[c] -= f * prow[c];
                }
            }
            // columns that still matter: A right of the panel, and the whole inverse half
            const size_t c_lo = k1;
            const size_t len = ld - c_lo;
            // ---- the K scaled pivot rows (as they are at their own step), privately ----
            for (size_t j = 0; j < kp; ++j) {
                double* sj = &S[j * ld];
                const size_t pp = (size_t)piv[j];
                std::copy(src + pp * ld + c_lo, src + pp * ld + ld, sj + c_lo);
                for (size_t i2 = 0; i2 < j; ++i2) {
                    const double f = F[pp * K + i2];
                    if (f != 0.0) row_sub_scaled(sj + c_lo, &S[i2 * ld] + c_lo, f, len, avx2);
                }
                row_div(sj + c_lo, dd[j], len, avx2);
            }
            // ---- own rows: all steps of the panel on every element, in step order ----
            for (size_t r = lo; r < hi; ++r) {
                double* x = dst + r * ld;
                std::copy(src + r * ld + c_lo, src + r * ld + ld, x + c_lo);
                for (size_t j = 0; j < kp; ++j) {
                    if ((int)r == piv[j]) {
                        row_div(x + c_lo, dd[j], len, avx2);
                    } else {
                        const double f = F[r * K + j];
                        if (f != 0.0) row_sub_scaled(x + c_lo, &S[j * ld] + c_lo, f, len, avx2);
                    }
                }
            }
        }
        if (trace && t == 0) tr_us[1] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count();
        const double* fin = buf[panel & 1].data();
        for (size_t p = lo; p < hi; ++p) {
            double s = 0.0;
            const double* ii = fin + p * ld + n;
            for (size_t j = 0; j < n; ++j) s += (-alpha * ii[j]) * g[j];  // element (i, j) of the inverse
            step[(size_t)iperm[p]] = s;
        }
    });
    if (trace)
        std::fprintf(stderr, "[solve] P=%d workers awake after %.0f us, panels done after %.0f us, returned after %.0f us\n", P, tr_us[0], tr_us[1],
                     std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count());
}


void glibc_rand_fill(uint32_t seed, int32_t* out, size_t count) {
    int32_t r[31];
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
        const long hi = r[i - 1] / 127773, lo = r[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (int32_t)word;
    }
    int f = 3, b = 0;
    auto next = [&]() {
        const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
        r[f] = (int32_t)v;
        f = (f + 1) % 31, b = (b + 1) % 31;
        return (int32_t)(v >> 1);
    };
    for (int i = 0; i < 310; ++i) (void)next();
    for (size_t i = 0; i < count; ++i) out[i] = next();
}

}  // namespace dmsa
