// window_setup.cpp — include/dmsa_window_setup.h: the host-side producers of the hot path's inputs.
// O(#poses + #IMU samples) double arithmetic, like the reference; product code, no oracle dependency.  The only per-point part,
// tformIdPerPoint, is a device kernel (dmsa_traj_tform_indices in dmsa_api.cpp).
#include "../../include/dmsa_window_setup.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "host_math.h"

using namespace dmsa;

namespace {

inline void put3(double* m, int k, Vec3 v) { m[3 * k] = v.x, m[3 * k + 1] = v.y, m[3 * k + 2] = v.z; }
inline Mat3 skew(Vec3 v) { return Mat3{{0.0, -v.z, v.y, v.z, 0.0, -v.x, -v.y, v.x, 0.0}}; }  // helpers.h:39-49
inline Mat3 scaled(double s, const Mat3& A) {
    Mat3 B;
    for (int i = 0; i < 9; ++i) B.a[i] = s * A.a[i];
    return B;
}
inline Mat3 scaled(const Mat3& A, double s) {
    Mat3 B;
    for (int i = 0; i < 9; ++i) B.a[i] = A.a[i] * s;
    return B;
}
inline Mat3 plus(const Mat3& A, const Mat3& B) {
    Mat3 C;
    for (int i = 0; i < 9; ++i) C.a[i] = A.a[i] + B.a[i];
    return C;
}
inline Mat3 minus(const Mat3& A, const Mat3& B) {
    Mat3 C;
    for (int i = 0; i < 9; ++i) C.a[i] = A.a[i] - B.a[i];
    return C;
}

// Eigen's VectorXd::LinSpaced(n, 0, high) (linspaced_op_impl, low = 0 so never flipped): i * step, the last entry exactly high
void lin_spaced(int n, double high, double* out) {
    const double step = n == 1 ? 0.0 : (high - 0.0) / (double)(n - 1);
    for (int i = 0; i < n; ++i) out[i] = 0.0 + (double)i * step;
    if (n > 1) out[n - 1] = high;
}

// ---- ImuPreintegration (ImuPreintegration.h:23-139) ----------------------------------------------------------------------------
struct Preintegrator {
    Vec3 dpos{0, 0, 0}, dvel{0, 0, 0};
    Mat3 drot = Mat3::identity();
    double cov[81];  // row-major 9 x 9, order (rot, vel, pos)
    Preintegrator() { reset(); }
    void reset() {
        dpos = {0, 0, 0}, dvel = {0, 0, 0}, drot = Mat3::identity();
        std::fill(cov, cov + 81, 0.0);
    }
    static Mat3 jacobian_r(Vec3 rot) {  // :35-47
        const double n = length(rot);
        const Mat3 S = skew(rot);
        if (n < 0.00001) return Mat3::identity();
        const Mat3 first = scaled((1.0 - std::cos(n)) / std::pow(n, 2), S);
        const Mat3 second = scaled((n - std::sin(n)) / std::pow(n, 3), S) * S;
        return plus(minus(Mat3::identity(), first), second);
    }
    void add(Vec3 omega, Vec3 acc, double dt, const double* gyr_cov /* col-major 3x3 */, const double* acc_cov) {  // :55-107
        const double dt2 = dt * dt;
        const Mat3 rot_incr = so3_exp(dt * omega);
        double A[81], B[54];  // 9 x 9 and 9 x 6, row-major
        std::fill(A, A + 81, 0.0), std::fill(B, B + 54, 0.0);
        for (int i = 0; i < 9; ++i) A[9 * i + i] = 1.0;
        auto setA = [&](int r0, int c0, const Mat3& M) {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) A[9 * (r0 + r) + c0 + c] = M(r, c);
        };
        auto setB = [&](int r0, int c0, const Mat3& M) {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) B[6 * (r0 + r) + c0 + c] = M(r, c);
        };
        const Mat3 Sa = skew(acc);
        setA(0, 0, transposed(rot_incr));
        setA(3, 0, scaled(scaled(-1.0, drot) * Sa, dt));
        setA(6, 0, scaled(scaled(-0.5, drot) * Sa, dt2));
        setA(6, 3, scaled(dt, Mat3::identity()));
        setB(0, 0, scaled(jacobian_r(so3_log(drot)), dt));
        setB(3, 3, scaled(drot, dt));
        setB(6, 3, scaled(scaled(0.5, drot), dt2));
        double N[36];
        std::fill(N, N + 36, 0.0);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) N[6 * r + c] = gyr_cov[3 * c + r], N[6 * (3 + r) + 3 + c] = acc_cov[3 * c + r];
        // cov = (A cov) A^T + (B N) B^T
        double AC[81], BN[54], out[81];
        for (int r = 0; r < 9; ++r)
            for (int c = 0; c < 9; ++c) {
                double s = 0.0;
                for (int k = 0; k < 9; ++k) s += A[9 * r + k] * cov[9 * k + c];
                AC[9 * r + c] = s;
            }
        for (int r = 0; r < 9; ++r)
            for (int c = 0; c < 6; ++c) {
                double s = 0.0;
                for (int k = 0; k < 6; ++k) s += B[6 * r + k] * N[6 * k + c];
                BN[6 * r + c] = s;
            }
        for (int r = 0; r < 9; ++r)
            for (int c = 0; c < 9; ++c) {
                double s1 = 0.0, s2 = 0.0;
                for (int k = 0; k < 9; ++k) s1 += AC[9 * r + k] * A[9 * c + k];
                for (int k = 0; k < 6; ++k) s2 += BN[6 * r + k] * B[6 * c + k];
                out[9 * r + c] = s1 + s2;
            }
        std::memcpy(cov, out, sizeof(out));
        const Vec3 half_ra = scaled(0.5, drot) * acc;
        dpos = dpos + ((dt * dvel) + (dt2 * half_ra));
        dvel = dvel + (dt * (drot * acc));
        drot = drot * rot_incr;
    }
};

// Matrix<double,9,9>::inverse(): partial-pivot elimination on [A | I] (the same statement as the explicit H^-1 of the LM step)
void invert_rowmajor(const double* Ain, int n, double* inv) {
    std::vector<double> A(Ain, Ain + (size_t)n * n);
    std::fill(inv, inv + (size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = 1.0;
    for (int c0 = 0; c0 < n; ++c0) {
        int piv = c0;
        double best = std::fabs(A[(size_t)c0 * n + c0]);
        for (int r = c0 + 1; r < n; ++r)
            if (std::fabs(A[(size_t)r * n + c0]) > best) best = std::fabs(A[(size_t)r * n + c0]), piv = r;
        if (piv != c0) {
            std::swap_ranges(&A[(size_t)c0 * n], &A[(size_t)c0 * n] + n, &A[(size_t)piv * n]);
            std::swap_ranges(&inv[(size_t)c0 * n], &inv[(size_t)c0 * n] + n, &inv[(size_t)piv * n]);
        }
        const double d = A[(size_t)c0 * n + c0];
        for (int c = 0; c < n; ++c) A[(size_t)c0 * n + c] /= d, inv[(size_t)c0 * n + c] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == c0) continue;
            const double f = A[(size_t)r * n + c0];
            if (f == 0.0) continue;
            for (int c = 0; c < n; ++c) A[(size_t)r * n + c] -= f * A[(size_t)c0 * n + c], inv[(size_t)r * n + c] -= f * inv[(size_t)c0 * n + c];
        }
    }
}

// chain helpers on the caller's arrays (ConsecutivePoses.h:26-67)
void relative_to_global(int n, const double* rel_o, const double* rel_t, double* glob_o, double* glob_t) {
    PoseChain c;
    c.resize(n);
    std::copy(rel_o, rel_o + 3 * n, c.rel_o.begin()), std::copy(rel_t, rel_t + 3 * n, c.rel_t.begin());
    c.relative_to_global();
    std::copy(c.glob_o.begin(), c.glob_o.end(), glob_o), std::copy(c.glob_t.begin(), c.glob_t.end(), glob_t);
}
void global_to_relative(int n, const double* glob_o, const double* glob_t, double* rel_o, double* rel_t) {
    PoseChain c;
    c.resize(n);
    std::copy(glob_o, glob_o + 3 * n, c.glob_o.begin()), std::copy(glob_t, glob_t + 3 * n, c.glob_t.begin());
    c.global_to_relative();
    std::copy(c.rel_o.begin(), c.rel_o.end(), rel_o), std::copy(c.rel_t.begin(), c.rel_t.end(), rel_t);
}

// getInterpRotation (ContinuousTrajectory.h:570-591)
Vec3 interp_rotation(const double* glob_o, const double* stamps, int n, double t) {
    const int right = (int)(std::lower_bound(stamps, stamps + n - 1, t) - stamps);
    if (right == 0) return col3(glob_o, 0);
    const double t_rel = (t - stamps[right - 1]) / (stamps[right] - stamps[right - 1]);
    return slerp_axang(col3(glob_o, right - 1), col3(glob_o, right), t_rel);
}

// boost::math::barycentric_rational<double>::prime (Boost 1.71 barycentric_rational_detail.hpp)
double fh_prime(const FloaterHormann2& fh, const double* y, double x) {
    const double rx = fh.eval(y, x);
    const size_t n = fh.x.size();
    double numerator = 0.0, denominator = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (x == fh.x[i]) {
            double sum = 0.0;
            for (size_t j = 0; j < n; ++j) {
                if (j == i) continue;
                sum += fh.w[j] * (y[i] - y[j]) / (fh.x[i] - fh.x[j]);
            }
            return -sum / fh.w[i];
        }
        const double t = fh.w[i] / (x - fh.x[i]);
        const double diff = (rx - y[i]) / (x - fh.x[i]);
        numerator += t * diff;
        denominator += t;
    }
    return numerator / denominator;
}

// Eigen's AngleAxisd(Matrix3d) -> angle * axis: Quaterniond(mat) (trace branch / largest-diagonal branch), then AngleAxis(q)
Vec3 angle_axis_of(const Mat3& m) {
    double q[4];  // x, y, z, w
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t, q[1] = (m(0, 2) - m(2, 0)) * t, q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m(k, j) - m(j, k)) * t;
        q[j] = (m(j, i) + m(i, j)) * t;
        q[k] = (m(k, i) + m(i, k)) * t;
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n == 0.0) return {0.0, 0.0, 0.0};  // angle 0 (axis (1,0,0))
    const double angle = 2.0 * std::atan2(n, std::fabs(q[3]));
    if (q[3] < 0.0) n = -n;
    return {angle * (q[0] / n), angle * (q[1] / n), angle * (q[2] / n)};
}

// initGravityDir (ContinuousTrajectory.h:263-299)
void init_gravity_dir(dmsa_traj_state* cur) {
    const Vec3 measured = col3(cur->acc_meas, 0);
    const Vec3 v1{cur->gravity[0], cur->gravity[1], cur->gravity[2]};
    const Vec3 v2 = -1.0 * measured;
    Vec3 axis{v1.y * v2.z - v1.z * v2.y, v1.z * v2.x - v1.x * v2.z, v1.x * v2.y - v1.y * v2.x};
    const double sq = axis.x * axis.x + axis.y * axis.y + axis.z * axis.z;
    if (sq > 0.0) {  // Eigen normalized(): a zero vector stays zero
        const double nrm = std::sqrt(sq);
        axis = {axis.x / nrm, axis.y / nrm, axis.z / nrm};
    }
    const double angle = std::acos((v1.x * v2.x + v1.y * v2.y + v1.z * v2.z) / (length(v1) * length(v2)));
    const Mat3 K = skew(axis);
    const Mat3 R = plus(plus(Mat3::identity(), scaled(std::sin(angle), K)), scaled(1.0 - std::cos(angle), K) * K);
    put3(cur->rel_orient, 0, angle_axis_of(transposed(R)));
    relative_to_global(cur->num_control_poses, cur->rel_orient, cur->rel_transl, cur->glob_orient, cur->glob_transl);
}

// getImuIntegratedParams (ContinuousTrajectory.h:470-516)
void imu_integrate(const dmsa_traj_state* cur, double t0, Vec3 axang0, Vec3 pos0, Vec3 v0, double tend, Vec3& axang_end, Vec3& pos_end, Vec3& v_end) {
    const double* tt = cur->traj_time;
    const int n_total = cur->n_total;
    int index = (int)(std::lower_bound(tt, tt + n_total - 1, t0) - tt);
    // the reference reads trajTime[index - 1] even for index == 0; there t0 <= trajTime[0] = 0 makes |t0 - next| the smaller one
    const double prev = index > 0 ? tt[index - 1] : -std::numeric_limits<double>::infinity();
    const double next = tt[index];
    if (std::fabs(t0 - prev) < std::fabs(t0 - next)) index = index - 1;
    Mat3 R = so3_exp(axang0);
    Vec3 pos_w = pos0, vel_w = v0;
    const double dt = cur->dt_res, dt2 = dt * dt;
    const Vec3 g{cur->gravity[0], cur->gravity[1], cur->gravity[2]};
    double curr = t0;
    while (std::fabs(curr + dt - tend) < std::fabs(curr - tend) && index < n_total) {
        const Vec3 acc = col3(cur->acc_meas, index);
        pos_w = ((pos_w + (dt * vel_w)) + (dt2 * (0.5 * g))) + (dt2 * (scaled(0.5, R) * acc));
        vel_w = (vel_w + (dt * g)) + (dt * (R * acc));
        R = R * so3_exp(dt * col3(cur->ang_vel_meas, index));
        index += 1;
        curr += dt;
    }
    axang_end = so3_log(R), pos_end = pos_w, v_end = vel_w;
}

}  // namespace

// ---- ImuBuffer (ImuBuffer.h:14-175) --------------------------------------------------------------------------------------------
struct dmsa_imu_buffer {
    std::vector<double> acc, ang_vel, stamps;  // 3 x max, 3 x max, max
    Vec3 bias_gyr{0, 0, 0};
    int oldest_index = 0, max_num_meas = 10000, num_updates = 0;
};

extern "C" {

int dmsa_imu_buffer_create(int32_t max_num_meas, dmsa_imu_buffer** out) {
    if (!out || max_num_meas < 2) return DMSA_ERR_INVALID;
    dmsa_imu_buffer* b = new (std::nothrow) dmsa_imu_buffer();
    if (!b) return DMSA_ERR_NOMEM;
    b->max_num_meas = max_num_meas;
    b->acc.assign(3 * (size_t)max_num_meas, 0.0), b->ang_vel.assign(3 * (size_t)max_num_meas, 0.0), b->stamps.assign((size_t)max_num_meas, 0.0);
    *out = b;
    return DMSA_OK;
}
void dmsa_imu_buffer_destroy(dmsa_imu_buffer* b) { delete b; }

int dmsa_imu_buffer_add(dmsa_imu_buffer* b, const double acc[3], const double ang_vel[3], double stamp) {
    if (!b || !acc || !ang_vel) return DMSA_ERR_INVALID;
    const int at = b->oldest_index;
    put3(b->acc.data(), at, {acc[0], acc[1], acc[2]});
    put3(b->ang_vel.data(), at, {ang_vel[0] - b->bias_gyr.x, ang_vel[1] - b->bias_gyr.y, ang_vel[2] - b->bias_gyr.z});
    b->stamps[(size_t)at] = stamp;
    if (++b->oldest_index == b->max_num_meas) b->oldest_index = 0;
    ++b->num_updates;
    if (b->num_updates == 50) {  // :60-64 rowwise().mean() over the first numUpdates columns
        const int n = std::min(b->num_updates, b->max_num_meas);
        Vec3 s{0, 0, 0};
        for (int k = 0; k < n; ++k) s = s + col3(b->ang_vel.data(), k);
        b->bias_gyr = {s.x / (double)n, s.y / (double)n, s.z / (double)n};
    }
    return DMSA_OK;
}

int dmsa_imu_buffer_closest(const dmsa_imu_buffer* b, double t, double acc_out[3], double ang_vel_out[3], double* timediff_out) {
    if (!b || b->num_updates <= 0) return DMSA_ERR_INVALID;
    const double* S = b->stamps.data();
    int index;
    double diff;
    if (b->num_updates <= b->max_num_meas || b->oldest_index == 0) {  // :72-85
        const int end = std::min(b->max_num_meas - 1, b->num_updates - 1);
        index = (int)(std::lower_bound(S, S + end, t) - S);
        diff = std::fabs(t - S[index]);
    } else {  // :86-122: the two halves of the wrapped buffer, each without its last slot
        const int right = (int)(std::lower_bound(S + b->oldest_index, S + b->max_num_meas - 1, t) - S);
        const int left = (int)(std::lower_bound(S, S + b->oldest_index - 1, t) - S);
        index = std::fabs(t - S[right]) < std::fabs(t - S[left]) ? right : left;
        diff = t - S[index];
    }
    if (acc_out) std::copy(&b->acc[3 * (size_t)index], &b->acc[3 * (size_t)index] + 3, acc_out);
    if (ang_vel_out) std::copy(&b->ang_vel[3 * (size_t)index], &b->ang_vel[3 * (size_t)index] + 3, ang_vel_out);
    if (timediff_out) *timediff_out = diff;
    return DMSA_OK;
}

int dmsa_imu_buffer_state(const dmsa_imu_buffer* b, int32_t* num_updates, int32_t* oldest_index, double bias_gyr[3], double* latest_stamp,
                          double* oldest_stamp) {
    if (!b) return DMSA_ERR_INVALID;
    if (num_updates) *num_updates = b->num_updates;
    if (oldest_index) *oldest_index = b->oldest_index;
    if (bias_gyr) bias_gyr[0] = b->bias_gyr.x, bias_gyr[1] = b->bias_gyr.y, bias_gyr[2] = b->bias_gyr.z;
    if (latest_stamp)  // :127-135
        *latest_stamp = b->num_updates == 0 ? -1.0 : (b->oldest_index == 0 ? b->stamps[(size_t)b->max_num_meas - 1] : b->stamps[(size_t)b->oldest_index - 1]);
    if (oldest_stamp)  // :137-145
        *oldest_stamp = b->num_updates == 0 ? -1.0 : (b->num_updates < b->max_num_meas ? b->stamps[0] : b->stamps[(size_t)b->oldest_index]);
    return DMSA_OK;
}

// ---- initTraj --------------------------------------------------------------------------------------------------------------------
int dmsa_traj_dims(double t_min, double t_max, double dt_res, double* horizon_out, int32_t* n_total_out) {
    if (!(dt_res > 0.0) || !(t_max >= t_min) || !horizon_out || !n_total_out) return DMSA_ERR_INVALID;
    const double horizon = t_max - t_min + dt_res;           // :307
    const double n = std::round(horizon / dt_res) + 1;        // :308
    if (!(n < 2147483647.0)) return DMSA_ERR_INVALID;
    *horizon_out = horizon, *n_total_out = (int32_t)n;
    return DMSA_OK;
}

int dmsa_traj_grids(double horizon, double dt_res, int32_t n_total, int32_t C, double* traj_time_out, double* stamps_out, int32_t* param_indices_out) {
    if (n_total < 2 || C < 2 || !(dt_res > 0.0) || !traj_time_out || !stamps_out) return DMSA_ERR_INVALID;
    lin_spaced(n_total, horizon, traj_time_out);  // :322
    lin_spaced(C, horizon, stamps_out);           // :331
    if (param_indices_out)
        for (int k = 0; k < C; ++k) param_indices_out[k] = (int32_t)std::round(stamps_out[k] / dt_res);  // :334-335
    return DMSA_OK;
}

// ---- transferImuMeasurements -------------------------------------------------------------------------------------------------------
int dmsa_traj_transfer_imu(const dmsa_imu_buffer* b, double t0, const double* traj_time, int32_t n_total, double* acc_meas_out, double* ang_vel_meas_out,
                           double* worst_timediff_out) {
    if (!b || !traj_time || n_total < 1 || !acc_meas_out || !ang_vel_meas_out) return DMSA_ERR_INVALID;
    double worst = 0.0;
    for (int k = 0; k < n_total; ++k) {
        double diff = 0.0;
        const int rc = dmsa_imu_buffer_closest(b, t0 + traj_time[k], acc_meas_out + 3 * (size_t)k, ang_vel_meas_out + 3 * (size_t)k, &diff);
        if (rc != DMSA_OK) return rc;
        worst = std::max(worst, std::fabs(diff));
    }
    if (worst_timediff_out) *worst_timediff_out = worst;
    return DMSA_OK;
}

// ---- updatePreintFactors -------------------------------------------------------------------------------------------------------------
int dmsa_traj_preint_factors(int32_t n_total, int32_t C, const int32_t* param_indices, double dt_res, const double* acc_meas, const double* ang_vel_meas,
                             const double gyr_cov[9], const double acc_cov[9], double* preint_rot_out, double* preint_pos_out, double* preint_vel_out,
                             double* cov_pvrot_inv_out, double preint_pos_horizon_out[3]) {
    if (n_total < 1 || C < 2 || !param_indices || !acc_meas || !ang_vel_meas || !gyr_cov || !acc_cov || !preint_rot_out || !preint_pos_out || !preint_vel_out ||
        !cov_pvrot_inv_out)
        return DMSA_ERR_INVALID;
    for (int k = 0; k < C; ++k)
        if (param_indices[k] < 0 || param_indices[k] > n_total) return DMSA_ERR_INVALID;
    static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::copy(I3, I3 + 9, preint_rot_out);  // :522-524
    std::fill(preint_pos_out, preint_pos_out + 3, 0.0), std::fill(preint_vel_out, preint_vel_out + 3, 0.0);
    std::fill(cov_pvrot_inv_out, cov_pvrot_inv_out + 81, 0.0);
    Preintegrator pre;
    for (int k = 1; k < C; ++k) {
        pre.reset();
        for (int t = param_indices[k - 1]; t < param_indices[k]; ++t) pre.add(col3(ang_vel_meas, t), col3(acc_meas, t), dt_res, gyr_cov, acc_cov);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) preint_rot_out[9 * (size_t)k + 3 * c + r] = pre.drot(r, c);
        put3(preint_pos_out, k, pre.dpos), put3(preint_vel_out, k, pre.dvel);
        double inv[81];
        invert_rowmajor(pre.cov, 9, inv);
        for (int r = 0; r < 9; ++r)
            for (int c = 0; c < 9; ++c) cov_pvrot_inv_out[81 * (size_t)k + 9 * c + r] = inv[9 * r + c];
    }
    if (preint_pos_horizon_out) {  // :557-567
        pre.reset();
        for (int t = 0; t < n_total; ++t) pre.add(col3(ang_vel_meas, t), col3(acc_meas, t), dt_res, gyr_cov, acc_cov);
        preint_pos_horizon_out[0] = pre.dpos.x, preint_pos_horizon_out[1] = pre.dpos.y, preint_pos_horizon_out[2] = pre.dpos.z;
    }
    return DMSA_OK;
}

// ---- updateInitialGuess ----------------------------------------------------------------------------------------------------------------
int dmsa_traj_update_initial_guess(int32_t* is_initialized, dmsa_traj_state* cur, dmsa_traj_state* old_traj, int32_t use_imu) {
    if (!is_initialized || !cur || cur->num_control_poses < 2 || !cur->stamps || !cur->rel_orient || !cur->rel_transl || !cur->glob_orient || !cur->glob_transl)
        return DMSA_ERR_INVALID;
    if (use_imu && (!cur->acc_meas || !cur->ang_vel_meas || !cur->traj_time || cur->n_total < 2)) return DMSA_ERR_INVALID;
    const int C = cur->num_control_poses;
    if (!*is_initialized) {  // :370-379
        if (use_imu) init_gravity_dir(cur);
        *is_initialized = 1;
        return DMSA_OK;
    }
    dmsa_traj_state* old = old_traj;
    if (!old || old->num_control_poses < 3 || !old->stamps || !old->rel_orient || !old->rel_transl || !old->glob_orient || !old->glob_transl) return DMSA_ERR_INVALID;
    const int Co = old->num_control_poses;
    relative_to_global(Co, old->rel_orient, old->rel_transl, old->glob_orient, old->glob_transl);  // :382
    int last_known = 0;
    for (int k = 0; k < C; ++k)
        if (cur->t0 + cur->stamps[k] < old->t0 + old->horizon) last_known = k;  // :384-388
    for (int k = 0; k <= last_known; ++k)  // :391-394
        put3(cur->glob_orient, k, interp_rotation(old->glob_orient, old->stamps, Co, cur->stamps[k] + cur->t0 - old->t0));
    FloaterHormann2 fh;
    if (!fh.build(old->stamps, Co)) return DMSA_ERR_INVALID;  // boost throws std::logic_error on coincident stamps
    double v0a[3];
    std::vector<double> y((size_t)Co);
    for (int a = 0; a < 3; ++a) {  // :399-419
        for (int j = 0; j < Co; ++j) y[(size_t)j] = old->glob_transl[3 * j + a];
        for (int j = 0; j <= last_known; ++j) cur->glob_transl[3 * j + a] = fh.eval(y.data(), cur->stamps[j] + cur->t0 - old->t0);
        v0a[a] = fh_prime(fh, y.data(), cur->stamps[last_known] + cur->t0 - old->t0);
    }
    global_to_relative(C, cur->glob_orient, cur->glob_transl, cur->rel_orient, cur->rel_transl);  // :422
    if (use_imu) {  // :424-452
        Vec3 pos0 = col3(cur->glob_transl, last_known), axang0 = col3(cur->glob_orient, last_known), v0{v0a[0], v0a[1], v0a[2]};
        for (int k = last_known; k < C - 1; ++k) {
            Vec3 axang_end, pos_end, v_end;
            imu_integrate(cur, cur->stamps[k], axang0, pos0, v0, cur->stamps[k + 1], axang_end, pos_end, v_end);
            put3(cur->glob_orient, k + 1, axang_end), put3(cur->glob_transl, k + 1, pos_end);
            axang0 = axang_end, pos0 = pos_end, v0 = v_end;
        }
        global_to_relative(C, cur->glob_orient, cur->glob_transl, cur->rel_orient, cur->rel_transl);
    } else {  // :453-466 constant relative motion (for last_known == 0 that is pose 0 itself, as in the reference)
        for (int k = last_known; k < C - 1; ++k) {
            put3(cur->rel_orient, k + 1, col3(cur->rel_orient, last_known));
            put3(cur->rel_transl, k + 1, col3(cur->rel_transl, last_known));
        }
        relative_to_global(C, cur->rel_orient, cur->rel_transl, cur->glob_orient, cur->glob_transl);
    }
    return DMSA_OK;
}

// ---- getSubmapGravityEstimate -------------------------------------------------------------------------------------------------------------
int dmsa_traj_submap_gravity_estimate(const dmsa_traj_state* s, const double preint_pos_horizon[3], double gravity_imu_out[3]) {
    if (!s || !preint_pos_horizon || !gravity_imu_out || s->num_control_poses < 3 || s->n_total < 2 || !s->stamps || !s->traj_time || !s->glob_orient || !s->glob_transl)
        return DMSA_ERR_INVALID;
    const int C = s->num_control_poses;
    FloaterHormann2 fh;
    if (!fh.build(s->stamps, C)) return DMSA_ERR_INVALID;
    std::vector<double> y((size_t)C);
    double d0[3], d1[3];
    for (int a = 0; a < 3; ++a) {  // denseGlobalPoses.Translations.col(0 / 1): the interpolant of updateTrajDenseTforms at trajTime 0 / 1
        for (int j = 0; j < C; ++j) y[(size_t)j] = s->glob_transl[3 * j + a];
        d0[a] = fh.eval(y.data(), s->traj_time[0]), d1[a] = fh.eval(y.data(), s->traj_time[1]);
    }
    const double one_div_t_res = 1.0 / s->dt_res;
    const Vec3 v_start_w = one_div_t_res * Vec3{d1[0] - d0[0], d1[1] - d0[1], d1[2] - d0[2]};
    const Mat3 Rt = transposed(so3_exp(col3(s->glob_orient, 0)));
    const Vec3 first = col3(s->glob_transl, 0), last = col3(s->glob_transl, C - 1);
    const Vec3 inner = (last - first) - (s->horizon * v_start_w);
    const Vec3 r = Rt * inner;
    const double denom = 0.5 * std::pow(s->horizon, 2);
    gravity_imu_out[0] = (r.x - preint_pos_horizon[0]) / denom, gravity_imu_out[1] = (r.y - preint_pos_horizon[1]) / denom,
    gravity_imu_out[2] = (r.z - preint_pos_horizon[2]) / denom;
    return DMSA_OK;
}

}  // extern "C"
