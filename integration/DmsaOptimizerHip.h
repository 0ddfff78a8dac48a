// DmsaOptimizerHip.h — reference-side binding of libdmsa_hip.so (copy to include/DMSA/ of davidskdds/DMSA_LiDAR_SLAM).
//
// Keeps the class name pattern and the optimizeSet signatures of DmsaOptimizer<PointT> (include/DMSA/DmsaOptimizer.h:54), extracts the
// state the OptimizablePointSet virtuals (OptimizablePointSet.h:18-56) would read from the two concrete models, and calls the C ABI
// (include/dmsa_hip.h; the points through include/dmsa_aos.h: the PCL containers as they lie in memory).  Needs Eigen / PCL like the rest of the reference, so it cannot be compiled in the graft image;
// scripts/build_ref_oracle.sh documents the environment.  The library has one summation order: the reference's.
#pragma once
#include <algorithm>
#include <stdexcept>
#include <vector>
#include <cstddef>
#include <dmsa_hip.h>
#include <dmsa_aos.h>
#include "ContinuousTrajectory.h"
#include "MapManagement.h"
#include "DmsaOptimizer.h"          // DmsaOptimSettings

template <typename PointT>
class DmsaOptimizerHip {
    dmsa_ctx* ctx_ = nullptr;
public:
    explicit DmsaOptimizerHip(int device = 0, unsigned flags = 0) {
        if (dmsa_create(device, flags, &ctx_) != DMSA_OK) throw std::runtime_error("no usable HIP device");
    }
    ~DmsaOptimizerHip() { dmsa_destroy(ctx_); }

    static dmsa_settings convert(const DmsaOptimSettings& s) {        // DmsaOptimizer.h:25-39, field for field
        return {s.num_iter, s.epsilon, s.use_analytic_jacobi, s.step_length_optim, s.max_step, s.gauss_split,
                s.grid_size_1_factor, s.grid_size_2_factor, s.min_num_points_per_set, s.min_num_gaussians,
                s.lambda_diag, s.use_centralization};
    }

    // == DmsaOptimizer<PointStampId>::optimizeSet(ContinuousTrajectory&, settings)
    // The clouds are handed over as they lie in memory (include/dmsa_aos.h): no per-point repacking on the host.
    void optimizeSet(ContinuousTrajectory& t, DmsaOptimSettings settings = DmsaOptimSettings()) {
        std::vector<dmsa_aos_view> scans;
        size_t N = 0;
        for (int pc = 0; pc < t.regPcBuffer->getNumElements(); ++pc) {             // updateGlobalPoints order (:137-155)
            auto& c = t.regPcBuffer->at(pc);
            scans.push_back({c.points.data(), (int64_t)c.points.size(), (int32_t)sizeof(PointStampId), (int32_t)offsetof(PointStampId, data),
                             (int32_t)offsetof(PointStampId, id), t.tformIdPerPoint[pc].data()});
            N += c.points.size();
        }
        const dmsa_aos_view stat{t.globalPoints.points.data() + N, (int64_t)(t.globalPoints.points.size() - N), (int32_t)sizeof(PointStampId),   // addStaticPoints tail (:158-172)
                                 (int32_t)offsetof(PointStampId, data), (int32_t)offsetof(PointStampId, id), nullptr};
        dmsa_window_problem p{};
        p.num_control_poses = t.controlPoses.numPoses;
        p.rel_orient = t.controlPoses.relativePoses.Orientations.data();
        p.rel_transl = t.controlPoses.relativePoses.Translations.data();
        p.stamps = t.controlPoses.stamps.data();
        p.n_total = t.n_total;             p.traj_time = t.trajTime.data();
        p.min_grid_size = t.minGridSize;
        p.use_imu = t.useImuErrorTerms;    p.dt_res = t.dt_res;  p.balancing_imu = t.balancingImu;
        std::copy(t.gravity.data(), t.gravity.data() + 3, p.gravity);
        std::vector<double> rot, pos, vel, cov;                                     // vector<Matrix..> -> flat col-major
        if (t.useImuErrorTerms) {
            for (auto& m : t.preintImuRots) rot.insert(rot.end(), m.data(), m.data() + 9);
            for (auto& v : t.preintRelPositions) pos.insert(pos.end(), v.data(), v.data() + 3);
            for (auto& v : t.preintRelVelocity) vel.insert(vel.end(), v.data(), v.data() + 3);
            for (auto& m : t.CovPVRot_inv) cov.insert(cov.end(), m.data(), m.data() + 81);
            p.param_indices = t.paramIndices.data();
            p.preint_rot = rot.data(); p.preint_pos = pos.data(); p.preint_vel = vel.data(); p.cov_pvrot_inv = cov.data();
        }
        dmsa_settings s = convert(settings); dmsa_report rep{};
        if (dmsa_optimize_window_aos(ctx_, &p, scans.data(), (int32_t)scans.size(), &stat, &s, &rep) != DMSA_OK) throw std::runtime_error(dmsa_last_error(ctx_));
        t.controlPoses.relative2global();                                           // poses were updated in place
        // final updateGlobalPoints (:149): x, y, z written into globalPoints, every other field of a point left alone
        dmsa_get_global_points_aos(ctx_, t.globalPoints.points.data(), (int64_t)t.globalPoints.points.size(), (int32_t)sizeof(PointStampId),
                                   (int32_t)offsetof(PointStampId, data), -1);
    }

    // == DmsaOptimizer<PointT>::adaptiveStepSize(set, params, step, error0) (DmsaOptimizer.h:152-182; public, called by optimizeSet only): on the
    // problem that the last optimizeSet left resident and its current Gaussians.  `params` becomes the best of the nine trials if it beats error0.
    int adaptiveStepSize(Eigen::VectorXd& params, const Eigen::VectorXd& step, double error0) {
        int32_t k = 0;
        if (dmsa_adaptive_step_size(ctx_, params.data(), step.data(), error0, &k) != DMSA_OK) throw std::runtime_error(dmsa_last_error(ctx_));
        return k;
    }

    // == DmsaOptimizer<PointNormal>::optimizeSet(MapManagement&, settings)
    void optimizeSet(MapManagement& m, DmsaOptimSettings settings = DmsaOptimSettings()) {
        const int F = m.keyframeDataBuffer.getNumElements();
        std::vector<dmsa_aos_view> frames; std::vector<double> grav, otr, orm; std::vector<int32_t> plaus;
        int64_t first = 0;
        for (int k = 0; k < F; ++k) {
            auto& kd = m.keyframeDataBuffer.at(k);
            auto& pts = kd.pointCloudLocal->points;                                // pcl::PointNormal: data[4] | data_n[4] | curvature, 48 bytes
            frames.push_back({pts.data(), (int64_t)pts.size(), (int32_t)sizeof(pcl::PointNormal), (int32_t)offsetof(pcl::PointNormal, data),
                              (int32_t)offsetof(pcl::PointNormal, data_n), m.ringIds.data() + first});   // MapManagement::ringIds is flat over the frames
            first += (int64_t)pts.size();
            grav.insert(grav.end(), kd.measuredGravity.data(), kd.measuredGravity.data() + 3); plaus.push_back(kd.gravityPlausible);
            otr.insert(otr.end(), kd.relativeTransl.data(), kd.relativeTransl.data() + 3);
            orm.insert(orm.end(), kd.relativeOrientMat.data(), kd.relativeOrientMat.data() + 9);
        }
        dmsa_keyframe_problem p{};
        p.num_frames = F;
        p.rel_orient = m.keyframePoses.relativePoses.Orientations.data();
        p.rel_transl = m.keyframePoses.relativePoses.Translations.data();
        p.min_grid_size = m.minGridSize;
        p.use_gravity = m.useGravityErrorTerms; p.use_odometry = m.useOdometryErrorTerms;
        std::copy(m.gravity.data(), m.gravity.data() + 3, p.gravity);
        std::copy(m.Cov_grav_inv.data(), m.Cov_grav_inv.data() + 9, p.cov_grav_inv);
        p.balancing_grav = m.balancingFactorGrav; p.balancing_odom = m.balancingFactorOdom;
        p.measured_gravity = grav.data(); p.gravity_plausible = plaus.data();
        p.odom_rel_transl = otr.data(); p.odom_rel_orient_mat = orm.data();
        std::copy(m.odometryTranslCovInv.data(), m.odometryTranslCovInv.data() + 9, p.odom_transl_cov_inv);
        std::copy(m.odometryOrientCovInv.data(), m.odometryOrientCovInv.data() + 9, p.odom_orient_cov_inv);
        dmsa_settings s = convert(settings); dmsa_report rep{};
        if (dmsa_optimize_keyframes_aos(ctx_, &p, frames.data(), F, &s, &rep) != DMSA_OK) throw std::runtime_error(dmsa_last_error(ctx_));
        m.keyframePoses.relative2global();
        m.updateGlobalPoints();   // or dmsa_get_global_points_aos(ctx_, m.globalPoints.points.data(), n, sizeof(pcl::PointNormal), 0, offsetof(pcl::PointNormal, data_n))
    }
};
