// pose_math.h — double-precision SO(3) / pose-chain arithmetic shared by the host code AND the device kernels of the optimizeSet loop.
//
// One definition, compiled twice (g++-style host pass and gfx950 device pass, both -ffp-contract=off): every function is a fixed sequence
// of IEEE +, -, *, /, sqrt and the transcendental sequences of include/dmsa_detmath.h, so the device-resident loop (loop_kernels.hip)
// produces the bits the host chain (host_math.cpp) produces.
//   helpers.h:51-57  axang2rotm  -> so3_exp      helpers.h:59-65  rotm2axang -> so3_log
//   ConsecutivePoses.h:26-67     -> the per-pose steps of relative2global / global2relative (chain_*)
//   boost barycentric_rational   -> fh2_eval     (ContinuousTrajectory.h:214-217)
//   ContinuousTrajectory.h:603-663 updateImuError -> imu_row;  MapManagement.h:210-252 -> gravity_row / odometry_row
#pragma once

#include <cmath>
#include <cstddef>

#include "../../include/dmsa_detmath.h"

#if defined(__HIPCC__)
#define DMSA_HD __host__ __device__ inline
#else
#define DMSA_HD inline
#endif

namespace dmsa {

struct Vec3 {
    double x, y, z;
};
DMSA_HD Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
DMSA_HD Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
DMSA_HD Vec3 operator*(double s, Vec3 a) { return {s * a.x, s * a.y, s * a.z}; }
DMSA_HD double length(Vec3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

struct Mat3 {
    double a[9];  // row-major
    DMSA_HD double& operator()(int r, int c) { return a[3 * r + c]; }
    DMSA_HD double operator()(int r, int c) const { return a[3 * r + c]; }
    static DMSA_HD Mat3 identity() { return Mat3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
};
DMSA_HD Mat3 operator*(const Mat3& A, const Mat3& B) {
    Mat3 C;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C(r, c) = A(r, 0) * B(0, c) + A(r, 1) * B(1, c) + A(r, 2) * B(2, c);
    return C;
}
DMSA_HD Vec3 operator*(const Mat3& A, Vec3 v) {
    return {A(0, 0) * v.x + A(0, 1) * v.y + A(0, 2) * v.z, A(1, 0) * v.x + A(1, 1) * v.y + A(1, 2) * v.z,
            A(2, 0) * v.x + A(2, 1) * v.y + A(2, 2) * v.z};
}
DMSA_HD Mat3 transposed(const Mat3& A) {
    Mat3 T;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T(r, c) = A(c, r);
    return T;
}

// exp of a skew matrix in closed form (Rodrigues).  Identity below EPSILON_ROT = 1e-5 (helpers.h:18,53).
DMSA_HD Mat3 so3_exp(Vec3 w) {
    const double theta = length(w);
    if (theta < 0.00001) return Mat3::identity();
    const double s = dmsa_det::det_sin(theta) / theta;
    const double sh = dmsa_det::det_sin(0.5 * theta);
    const double c = 2.0 * sh * sh / (theta * theta);
    const double t2 = theta * theta;
    Mat3 R;
    R(0, 0) = 1.0 + c * (w.x * w.x - t2);
    R(1, 1) = 1.0 + c * (w.y * w.y - t2);
    R(2, 2) = 1.0 + c * (w.z * w.z - t2);
    R(0, 1) = c * w.x * w.y - s * w.z;
    R(1, 0) = c * w.x * w.y + s * w.z;
    R(0, 2) = c * w.x * w.z + s * w.y;
    R(2, 0) = c * w.x * w.z - s * w.y;
    R(1, 2) = c * w.y * w.z - s * w.x;
    R(2, 1) = c * w.y * w.z + s * w.x;
    return R;
}

// principal log of a rotation through its unit quaternion (largest-pivot extraction), angle in [0, pi].
DMSA_HD Vec3 so3_log(const Mat3& R) {
    const double tr = R(0, 0) + R(1, 1) + R(2, 2);
    double qw, qx, qy, qz;
    if (tr > 0.0) {
        const double s = sqrt(tr + 1.0) * 2.0;
        qw = 0.25 * s;
        qx = (R(2, 1) - R(1, 2)) / s;
        qy = (R(0, 2) - R(2, 0)) / s;
        qz = (R(1, 0) - R(0, 1)) / s;
    } else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) {
        const double s = sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2.0;
        qw = (R(2, 1) - R(1, 2)) / s;
        qx = 0.25 * s;
        qy = (R(0, 1) + R(1, 0)) / s;
        qz = (R(0, 2) + R(2, 0)) / s;
    } else if (R(1, 1) > R(2, 2)) {
        const double s = sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2.0;
        qw = (R(0, 2) - R(2, 0)) / s;
        qx = (R(0, 1) + R(1, 0)) / s;
        qy = 0.25 * s;
        qz = (R(1, 2) + R(2, 1)) / s;
    } else {
        const double s = sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2.0;
        qw = (R(1, 0) - R(0, 1)) / s;
        qx = (R(0, 2) + R(2, 0)) / s;
        qy = (R(1, 2) + R(2, 1)) / s;
        qz = 0.25 * s;
    }
    const double n = sqrt(qx * qx + qy * qy + qz * qz);
    if (n == 0.0) return {0.0, 0.0, 0.0};
    const double angle = 2.0 * dmsa_det::det_atan2(n, fabs(qw));
    const double k = angle / (qw < 0.0 ? -n : n);
    return {qx * k, qy * k, qz * k};
}

// pose k of a 3 x n column-major array
DMSA_HD Vec3 col3(const double* m, int k) { return {m[3 * k], m[3 * k + 1], m[3 * k + 2]}; }
DMSA_HD void set_col3(double* m, int k, Vec3 v) { m[3 * k] = v.x, m[3 * k + 1] = v.y, m[3 * k + 2] = v.z; }

// ---- ConsecutivePoses::relative2global (ConsecutivePoses.h:26-43), one pose per call: (R, T) is the running pose ----
DMSA_HD void chain_step(Mat3& R, Vec3& T, Vec3 rel_o_k, Vec3 rel_t_k, Vec3& glob_o_k, Vec3& glob_t_k) {
    T = T + R * rel_t_k;
    glob_t_k = T;
    R = R * so3_exp(rel_o_k);
    glob_o_k = so3_log(R);
}
DMSA_HD void chain_relative_to_global(int n, const double* rel_o, const double* rel_t, double* glob_o, double* glob_t) {
    Mat3 R = Mat3::identity();
    Vec3 T{0, 0, 0};
    for (int k = 0; k < n; ++k) {
        Vec3 go, gt;
        chain_step(R, T, col3(rel_o, k), col3(rel_t, k), go, gt);
        set_col3(glob_o, k, go), set_col3(glob_t, k, gt);
    }
}
// ---- global2relative (ConsecutivePoses.h:45-67), pose k > 0 ----
DMSA_HD void unchain_step(Vec3 glob_o_prev, Vec3 glob_t_prev, Vec3 glob_o_k, Vec3 glob_t_k, Vec3& rel_o_k, Vec3& rel_t_k) {
    const Mat3 R1t = transposed(so3_exp(glob_o_prev));
    const Mat3 R2 = so3_exp(glob_o_k);
    rel_o_k = so3_log(R1t * R2);
    rel_t_k = R1t * (glob_t_k - glob_t_prev);
}
DMSA_HD void chain_global_to_relative(int n, const double* glob_o, const double* glob_t, double* rel_o, double* rel_t) {
    set_col3(rel_o, 0, col3(glob_o, 0));
    set_col3(rel_t, 0, col3(glob_t, 0));
    for (int k = n - 1; k > 0; --k) {
        Vec3 ro, rt;
        unchain_step(col3(glob_o, k - 1), col3(glob_t, k - 1), col3(glob_o, k), col3(glob_t, k), ro, rt);
        set_col3(rel_o, k, ro), set_col3(rel_t, k, rt);
    }
}

// ---- Floater–Hormann (d = 2) evaluation with boost's exact-node short-circuit; y has stride `ys` doubles ----
DMSA_HD double fh2_eval(int n, const double* x, const double* w, const double* y, int ys, double t) {
    double num = 0.0, den = 0.0;
    for (int i = 0; i < n; ++i) {
        if (t == x[i]) return y[(size_t)i * ys];
        const double q = w[i] / (t - x[i]);
        num += q * y[(size_t)i * ys];
        den += q;
    }
    return num / den;
}

// ---- IMU factor row k (1 <= k < C) of updateImuError (ContinuousTrajectory.h:603-663), AFTER its global2relative ----
struct ImuConsts {
    int use_imu;
    double dt_res, balancing_imu;
    double gravity[3];
    const int* param_indices;     // C
    const double* preint_rot;     // C x 9 column-major
    const double* preint_pos;     // C x 3
    const double* preint_vel;     // C x 3
    const double* cov_inv;        // C x 81 column-major
};
// the twelve dense-trajectory values row k needs, e = 3 * point + axis with the points (i0 + 1, i0, i1, i1 - 1) in the order the row uses them:
// independent Floater-Hormann evaluations (the device computes them on twelve lanes, k_loop_chain)
DMSA_HD double imu_dense_value(int k, int e, int C, const double* stamps, const double* fhw, const double* traj_time, const ImuConsts& c, const double* glob_t) {
    const int i0 = c.param_indices[k - 1], i1 = c.param_indices[k];
    const int pt = e / 3, a = e - 3 * pt;
    const int j = pt == 0 ? i0 + 1 : pt == 1 ? i0 : pt == 2 ? i1 : i1 - 1;
    return fh2_eval(C, stamps, fhw, glob_t + a, 3, traj_time[j]);
}
// row k from its twelve dense values d[e]
DMSA_HD double imu_row_from_dense(int k, const double* d, const double* stamps, const ImuConsts& c, const double* glob_o, const double* glob_t, Vec3 rel_o_k) {
    const double inv_dt = 1.0 / c.dt_res;
    const Mat3 Rst = transposed(so3_exp(col3(glob_o, k - 1)));
    const double delta_t = stamps[k] - stamps[k - 1];
    const Vec3 v_start = inv_dt * (Vec3{d[0], d[1], d[2]} - Vec3{d[3], d[4], d[5]});
    const Vec3 v_end = inv_dt * (Vec3{d[6], d[7], d[8]} - Vec3{d[9], d[10], d[11]});
    const double half_dt2 = 0.5 * (delta_t * delta_t);  // std::pow(delta_t, 2): exact square, correctly rounded
    const Vec3 pk = col3(glob_t, k), pk1 = col3(glob_t, k - 1);
    const Vec3 tmp_p{pk.x - pk1.x - v_start.x * delta_t - half_dt2 * c.gravity[0], pk.y - pk1.y - v_start.y * delta_t - half_dt2 * c.gravity[1],
                     pk.z - pk1.z - v_start.z * delta_t - half_dt2 * c.gravity[2]};
    const Vec3 dp = Rst * tmp_p;
    Mat3 P;
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) P(r, cc) = c.preint_rot[9 * (size_t)k + 3 * cc + r];
    const Vec3 rot_err = so3_log(transposed(P) * so3_exp(rel_o_k));
    const Vec3 tmp_v{v_end.x - v_start.x - c.gravity[0] * delta_t, v_end.y - v_start.y - c.gravity[1] * delta_t, v_end.z - v_start.z - c.gravity[2] * delta_t};
    const Vec3 dv = Rst * tmp_v;
    const double ce[9] = {rot_err.x, rot_err.y, rot_err.z,
                          dv.x - c.preint_vel[3 * (size_t)k], dv.y - c.preint_vel[3 * (size_t)k + 1], dv.z - c.preint_vel[3 * (size_t)k + 2],
                          dp.x - c.preint_pos[3 * (size_t)k], dp.y - c.preint_pos[3 * (size_t)k + 1], dp.z - c.preint_pos[3 * (size_t)k + 2]};
    const double* Ci = c.cov_inv + 81 * (size_t)k;
    double q = 0.0, left[9];
    for (int j = 0; j < 9; ++j) {
        double s = 0.0;
        for (int i = 0; i < 9; ++i) s += ce[i] * Ci[9 * j + i];
        left[j] = s;
    }
    for (int j = 0; j < 9; ++j) q += left[j] * ce[j];
    q *= c.balancing_imu;
    return sqrt(q);
}
DMSA_HD double imu_row(int k, int C, const double* stamps, const double* fhw, const double* traj_time, const ImuConsts& c, const double* glob_o,
                       const double* glob_t, Vec3 rel_o_k) {
    double d[12];
    for (int e = 0; e < 12; ++e) d[e] = imu_dense_value(k, e, C, stamps, fhw, traj_time, c, glob_t);
    return imu_row_from_dense(k, d, stamps, c, glob_o, glob_t, rel_o_k);
}

// ---- gravity / odometry rows of the keyframe model (MapManagement.h:210-252) ----
struct KeyframeRowConsts {
    int use_gravity, use_odometry;
    double gravity[3];
    double cov_grav_inv[9], balancing_grav, balancing_odom;
    double odom_transl_cov_inv[9], odom_orient_cov_inv[9];
    const double* measured_gravity;   // F x 3
    const int* gravity_plausible;     // F
    const double* odom_transl;        // F x 3
    const double* odom_orient_mat;    // F x 9 column-major
};
DMSA_HD double quad_form3(Vec3 d, const double* Ci /* col-major */) {
    const double l0 = d.x * Ci[0] + d.y * Ci[1] + d.z * Ci[2];
    const double l1 = d.x * Ci[3] + d.y * Ci[4] + d.z * Ci[5];
    const double l2 = d.x * Ci[6] + d.y * Ci[7] + d.z * Ci[8];
    return l0 * d.x + l1 * d.y + l2 * d.z;
}
// row k of updateGravityErrors: exactly 0 for frame 0 and for implausible frames
DMSA_HD double gravity_row(int k, const KeyframeRowConsts& c, Vec3 glob_o_k) {
    if (k == 0 || !c.gravity_plausible[k]) return 0.0;
    const Vec3 m{c.measured_gravity[3 * (size_t)k], c.measured_gravity[3 * (size_t)k + 1], c.measured_gravity[3 * (size_t)k + 2]};
    Vec3 d = so3_exp(glob_o_k) * m;
    d = {d.x - c.gravity[0], d.y - c.gravity[1], d.z - c.gravity[2]};
    double q = quad_form3(d, c.cov_grav_inv);
    q *= c.balancing_grav;
    return sqrt(q);
}
// row k - 1 of updateOdometryErrors (1 <= k < F)
DMSA_HD double odometry_row(int k, const KeyframeRowConsts& c, Vec3 rel_o_k, Vec3 rel_t_k) {
    const Vec3 o{c.odom_transl[3 * (size_t)k], c.odom_transl[3 * (size_t)k + 1], c.odom_transl[3 * (size_t)k + 2]};
    const Vec3 td = o - rel_t_k;
    Mat3 Rm;
    for (int r = 0; r < 3; ++r)
        for (int cc = 0; cc < 3; ++cc) Rm(r, cc) = c.odom_orient_mat[9 * (size_t)k + 3 * cc + r];
    const Vec3 od = so3_log(transposed(so3_exp(rel_o_k)) * Rm);
    double q = 0.0;
    q += quad_form3(td, c.odom_transl_cov_inv);
    q += quad_form3(od, c.odom_orient_cov_inv);
    q *= c.balancing_odom;
    return sqrt(q);
}

}  // namespace dmsa
