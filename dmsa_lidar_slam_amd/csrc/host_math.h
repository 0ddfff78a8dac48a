// host_math.h — host-side double-precision pose math of the DMSA path (product code, no oracle dependency).
//
// What stays on the host: the control-pose chain, the parameter (de)vectorisation and the
// few additional error rows.  They are O(#poses) per evaluation; everything O(#points) lives in HIP kernels.
//   Poses / ConsecutivePoses      -> PoseChain          (Poses.h:64-76, ConsecutivePoses.h:26-67)
//   helpers.h:24-65               -> so3_exp / so3_log / slerp_axang
//   boost barycentric_rational    -> FloaterHormann2    (ContinuousTrajectory.h:214-217)
//   updateTrajDenseTforms         -> window_dense_table (ContinuousTrajectory.h:189-226), host variant
//   updateImuError                -> WindowHost::imu_rows          (ContinuousTrajectory.h:603-663)
//   updateGravityErrors/Odometry  -> KeyframeHost::additional_rows (MapManagement.h:162-252)
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

#include "../../include/dmsa_hip.h"
#include "pose_math.h"

namespace dmsa {

// Vec3 / Mat3, so3_exp (axang2rotm, helpers.h:51-57), so3_log (rotm2axang, helpers.h:59-65) and the per-pose chain steps live in
// pose_math.h: one definition for the host and for the device kernels of the loop
Vec3 slerp_axang(Vec3 a, Vec3 b, double t);   // helpers.h:24-37

// Global <-> relative pose chains.  Columns are stored contiguously: pose k = o[3k..3k+2], t[3k..3k+2]
// (identical to Eigen's 3xn column-major Matrix3Xd).
struct PoseChain {
    int n = 0;
    std::vector<double> rel_o, rel_t, glob_o, glob_t;
    void resize(int count);
    void relative_to_global();  // ConsecutivePoses.h:26-43
    void global_to_relative();  // ConsecutivePoses.h:45-67
    int num_params() const { return 6 * (n - 1); }
    void get_params(double* p) const;        // Poses.h:64-70
    void set_params(const double* p);        // Poses.h:72-76
};

// Floater–Hormann rational interpolant of order 2 through (x_i, y_i) — the weights depend on x only,
// so one weight set serves the three translation axes.
struct FloaterHormann2 {
    std::vector<double> x, w;
    bool build(const double* nodes, int n);   // false on coincident nodes (boost throws std::logic_error)
    double eval(const double* y, double t) const;
};

// Dense [R|t] table (n_total x 12 floats, row-major 3x4) from the GLOBAL control poses — host variant of the
// pose-table kernel; bit-reproducible against the CPU oracle (DMSA_FLAG_POSE_TABLE_HOST).
void window_dense_table(const PoseChain& ctrl, const std::vector<double>& stamps, const FloaterHormann2& fh,
                        const std::vector<double>& traj_time, float* table);
void keyframe_table(const PoseChain& frames, float* table);

// ---- host state of the two problem models -------------------------------------------------------
struct WindowHost {
    PoseChain ctrl;
    std::vector<double> stamps, traj_time;
    FloaterHormann2 fh;
    bool use_imu = false;
    double dt_res = 1e-3, balancing_imu = 1e-3;
    Vec3 gravity{0, 0, -9.805};
    std::vector<int32_t> param_indices;
    std::vector<double> preint_rot, preint_pos, preint_vel, cov_inv;  // as in dmsa_window_problem
    Vec3 origin{0, 0, 0};

    bool init(const dmsa_window_problem& p);
    int num_extra_rows() const { return use_imu ? ctrl.n - 1 : 0; }
    // updateImuError on the CURRENT chain state (runs global_to_relative first, like the reference)
    void imu_rows(double* rows);
    ImuConsts imu_consts() const;  // pointers into this object's vectors
};

struct KeyframeHost {
    PoseChain frames;
    bool use_gravity = false, use_odometry = false;
    Vec3 gravity{0, 0, -9.805};
    double cov_grav_inv[9], balancing_grav = 1.0, balancing_odom = 1000.0;
    std::vector<double> measured_gravity, odom_transl, odom_orient_mat;
    std::vector<int32_t> gravity_plausible;
    double odom_transl_cov_inv[9], odom_orient_cov_inv[9];

    bool init(const dmsa_keyframe_problem& p);
    int num_extra_rows() const;
    void additional_rows(double* rows) const;  // MapManagement.h:162-252, gravity rows then odometry rows
    KeyframeRowConsts row_consts() const;      // pointers into this object's vectors
};

// Dense symmetric solve for the LM step: step = -alpha * H^-1 * g with H^-1 from partial-pivot elimination
// (DmsaOptimizer.h:113, MatrixXd::inverse()).
// `par` (optional) runs a function on every thread of the caller's worker pool: fn(worker, num_workers); the row updates of a pivot
// step are spread over them for P >= 64 (same operations, same result).
using ParallelRun = std::function<void(const std::function<void(int, int)>&)>;
void lm_solve(const double* H /* PxP col-major */, const double* g, int P, double alpha, double* step, const ParallelRun* par = nullptr,
              int max_threads = 12 /* of par's workers that take part */);

// The first `count` values of glibc rand() after srand(seed): TYPE_3 additive feedback generator of random_r.c (degree 31,
// separation 3, 310 warm-up draws) -- randomGridDownsampling (helpers.h:86-94) draws one per octree leaf.
void glibc_rand_fill(uint32_t seed, int32_t* out, size_t count);

}  // namespace dmsa
