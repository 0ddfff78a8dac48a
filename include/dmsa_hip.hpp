// dmsa_hip.hpp — header-only C++ class surface over the C ABI (dmsa_hip.h), for hosts WITHOUT Eigen/PCL.
//
// The reference's seam is `DmsaOptimizer<PointT>::optimizeSet(OptimizablePointSet<PointT>&, DmsaOptimSettings)`
// (include/DMSA/DmsaOptimizer.h:54) on the two concrete sets `ContinuousTrajectory` (ContinuousTrajectory.h:24-669) and
// `MapManagement` (MapManagement.h:20-390).  This header keeps those names, the member names the hot path reads and the
// `optimizeSet` signature, with Eigen-layout-compatible plain containers (`Matrix3Xd` = 3 x n column-major doubles, PCL
// points = float[4]), so code written against the reference's classes reads the same.  A host that DOES have Eigen/PCL
// binds its real objects instead (`DmsaOptimizerHip`); both end in the same C calls.
//
// Error behaviour: the reference never throws on this path (early `break` + a message on std::cout,
// DmsaOptimizer.h:91,119,132,141); those exits are reported through `lastReport().stop_reason`.  Only conditions that cannot
// occur in the reference (no usable GPU, HIP failure, invalid buffers) throw `std::runtime_error` — there is no CPU fallback.
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "dmsa_hip.h"

namespace dmsa_hip {

// == Eigen::Matrix3Xd: data() is 3 x cols column-major
struct Matrix3Xd {
    std::vector<double> v;
    Matrix3Xd() = default;
    explicit Matrix3Xd(int cols) : v(3 * (size_t)cols, 0.0) {}
    int cols() const { return (int)(v.size() / 3); }
    double& operator()(int r, int c) { return v[3 * (size_t)c + r]; }
    double operator()(int r, int c) const { return v[3 * (size_t)c + r]; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
};

// == Poses (Poses.h:16-37)
struct Poses {
    Matrix3Xd Orientations, Translations;  // axis-angle | translation, one column per pose
};

// == DmsaOptimSettings (DmsaOptimizer.h:25-39), same names and defaults
struct DmsaOptimSettings {
    int num_iter = 15;
    double epsilon = 1e-5;
    bool use_analytic_jacobi = false;
    double step_length_optim = 0.05;
    double max_step = 0.01;
    bool gauss_split = false;
    float grid_size_1_factor = 2.0f;
    float grid_size_2_factor = 5.0f;
    int min_num_points_per_set = 6;
    int min_num_gaussians = 30;
    float lambda_diag = 1e-5f;
    bool use_centralization = true;

    dmsa_settings c() const {
        dmsa_settings s;
        s.num_iter = num_iter, s.epsilon = epsilon, s.use_analytic_jacobi = use_analytic_jacobi, s.step_length_optim = step_length_optim;
        s.max_step = max_step, s.gauss_split = gauss_split, s.grid_size_1_factor = grid_size_1_factor, s.grid_size_2_factor = grid_size_2_factor;
        s.min_num_points_per_set = min_num_points_per_set, s.min_num_gaussians = min_num_gaussians, s.lambda_diag = lambda_diag;
        s.use_centralization = use_centralization;
        return s;
    }
};

// == the state of a ContinuousTrajectory the hot path reads (names as in ContinuousTrajectory.h)
struct ContinuousTrajectory {
    struct {
        int numPoses = 0;
        Poses relativePoses;           // IN/OUT
        std::vector<double> stamps;    // strictly increasing
    } controlPoses;
    int n_total = 0;
    std::vector<double> trajTime;                  // n_total
    std::vector<std::array<float, 4>> localPoints;  // all clouds of regPcBuffer, chronological (updateGlobalPoints :137-155)
    std::vector<int32_t> tformIdPerPoint;           // flattened
    std::vector<int32_t> ringIds;                   // PointStampId::id
    std::vector<std::array<float, 4>> staticPoints; // addStaticPoints tail (:158-172), world frame
    std::vector<int32_t> staticRingIds;
    float minGridSize = 0.3f;
    bool useImuErrorTerms = false;
    double dt_res = 1e-3, balancingImu = 1e-3;
    std::array<double, 3> gravity{{0.0, 0.0, -9.805}};
    std::vector<int32_t> paramIndices;
    std::vector<double> preintImuRots, preintRelPositions, preintRelVelocity, CovPVRot_inv;  // flat, column-major per element
    std::vector<std::array<float, 4>> globalPoints;  // OUT: final updateGlobalPoints (DmsaOptimizer.h:149)
};

// == the state of a MapManagement (sub)map the hot path reads (names as in MapManagement.h / KeyframeData.h)
struct MapManagement {
    struct {
        Poses relativePoses;  // IN/OUT
    } keyframePoses;
    std::vector<int64_t> frameOffsets;               // F+1 prefix of points per keyframe
    std::vector<std::array<float, 4>> localPoints, localNormals;
    std::vector<int32_t> ringIds;
    float minGridSize = 0.3f;
    bool useGravityErrorTerms = false, useOdometryErrorTerms = false;
    std::array<double, 3> gravity{{0.0, 0.0, -9.805}};
    std::array<double, 9> Cov_grav_inv{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    double balancingFactorGrav = 1.0, balancingFactorOdom = 1000.0;
    std::vector<double> measuredGravity;      // F x 3
    std::vector<int32_t> gravityPlausible;    // F
    std::vector<double> relativeTransl;       // F x 3   (KeyframeData::relativeTransl)
    std::vector<double> relativeOrientMat;    // F x 9 column-major
    std::array<double, 9> odometryTranslCovInv{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, odometryOrientCovInv{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    std::vector<std::array<float, 4>> globalPoints;  // OUT
    int numFrames() const { return frameOffsets.empty() ? 0 : (int)frameOffsets.size() - 1; }
};

// == DmsaOptimizer<PointT> (DmsaOptimizer.h:41-150).  PointT is kept for source compatibility only.
template <typename PointT = void>
class DmsaOptimizer {
public:
    explicit DmsaOptimizer(int device = 0, unsigned flags = 0) {
        const int rc = dmsa_create(device, flags, &ctx_);
        if (rc != DMSA_OK) throw std::runtime_error(rc == DMSA_ERR_NO_DEVICE ? "dmsa_hip: no usable HIP device (there is no CPU fallback)" : "dmsa_hip: dmsa_create failed");
    }
    ~DmsaOptimizer() { dmsa_destroy(ctx_); }
    DmsaOptimizer(const DmsaOptimizer&) = delete;
    DmsaOptimizer& operator=(const DmsaOptimizer&) = delete;

    // == optimizeSet(ContinuousTrajectory&, settings)   call site DmsaSlam.h:166
    void optimizeSet(ContinuousTrajectory& t, DmsaOptimSettings settings = DmsaOptimSettings()) {
        dmsa_window_problem p{};
        p.num_control_poses = t.controlPoses.numPoses;
        p.rel_orient = t.controlPoses.relativePoses.Orientations.data();
        p.rel_transl = t.controlPoses.relativePoses.Translations.data();
        p.stamps = t.controlPoses.stamps.data();
        p.n_total = t.n_total, p.traj_time = t.trajTime.data();
        p.num_points = (int64_t)t.localPoints.size();
        p.xyz_local = t.localPoints.empty() ? nullptr : t.localPoints[0].data();
        p.tform_idx = t.tformIdPerPoint.data(), p.ring_id = t.ringIds.data();
        p.num_static = (int64_t)t.staticPoints.size();
        p.xyz_static = t.staticPoints.empty() ? nullptr : t.staticPoints[0].data();
        p.ring_id_static = t.staticRingIds.data();
        p.min_grid_size = t.minGridSize;
        p.use_imu = t.useImuErrorTerms, p.dt_res = t.dt_res, p.balancing_imu = t.balancingImu;
        for (int k = 0; k < 3; ++k) p.gravity[k] = t.gravity[k];
        if (t.useImuErrorTerms) {
            p.param_indices = t.paramIndices.data(), p.preint_rot = t.preintImuRots.data(), p.preint_pos = t.preintRelPositions.data();
            p.preint_vel = t.preintRelVelocity.data(), p.cov_pvrot_inv = t.CovPVRot_inv.data();
        }
        const dmsa_settings s = settings.c();
        check(dmsa_optimize_window(ctx_, &p, &s, &report_), "dmsa_optimize_window");
        fetch_global(t.globalPoints, t.localPoints.size() + t.staticPoints.size());
    }

    // == optimizeSet(MapManagement&, settings)          call site DmsaSlam.h:228
    void optimizeSet(MapManagement& m, DmsaOptimSettings settings = DmsaOptimSettings()) {
        dmsa_keyframe_problem p{};
        p.num_frames = m.numFrames();
        p.rel_orient = m.keyframePoses.relativePoses.Orientations.data();
        p.rel_transl = m.keyframePoses.relativePoses.Translations.data();
        p.frame_offset = m.frameOffsets.data();
        p.xyz_local = m.localPoints.empty() ? nullptr : m.localPoints[0].data();
        p.normal_local = m.localNormals.empty() ? nullptr : m.localNormals[0].data();
        p.ring_id = m.ringIds.data();
        p.min_grid_size = m.minGridSize;
        p.use_gravity = m.useGravityErrorTerms, p.use_odometry = m.useOdometryErrorTerms;
        for (int k = 0; k < 3; ++k) p.gravity[k] = m.gravity[k];
        for (int k = 0; k < 9; ++k)
            p.cov_grav_inv[k] = m.Cov_grav_inv[k], p.odom_transl_cov_inv[k] = m.odometryTranslCovInv[k], p.odom_orient_cov_inv[k] = m.odometryOrientCovInv[k];
        p.balancing_grav = m.balancingFactorGrav, p.balancing_odom = m.balancingFactorOdom;
        p.measured_gravity = m.measuredGravity.data(), p.gravity_plausible = m.gravityPlausible.data();
        p.odom_rel_transl = m.relativeTransl.data(), p.odom_rel_orient_mat = m.relativeOrientMat.data();
        const dmsa_settings s = settings.c();
        check(dmsa_optimize_keyframes(ctx_, &p, &s, &report_), "dmsa_optimize_keyframes");
        fetch_global(m.globalPoints, m.localPoints.size());
    }

    const dmsa_report& lastReport() const { return report_; }
    dmsa_ctx* context() { return ctx_; }

private:
    void check(int rc, const char* what) {
        if (rc != DMSA_OK) throw std::runtime_error(std::string("dmsa_hip: ") + what + ": " + dmsa_last_error(ctx_));
    }
    void fetch_global(std::vector<std::array<float, 4>>& out, size_t n) {
        out.resize(n);
        if (n) check(dmsa_get_global_points(ctx_, out[0].data(), (int64_t)n), "dmsa_get_global_points");
    }
    dmsa_ctx* ctx_ = nullptr;
    dmsa_report report_{};
};

}  // namespace dmsa_hip
