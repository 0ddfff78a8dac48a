/*
 * dmsa_oracle.h — C API of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C++ restatement of the reference's
 * DMSA inner loop; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  Nothing in dmsa_lidar_slam_amd/ links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests/golden vectors for this path and its
 * third-party arithmetic (PCL 1.10 OctreePointCloud, Eigen 3.4, Boost 1.71
 * barycentric_rational) is not in /root/reference nor in this image, so it cannot be
 * compiled here.  Those pieces are restated from their published algorithms (see
 * SURVEY.md Appendix A) and cross-checked against numpy/scipy in tests/.
 */
#ifndef DMSA_ORACLE_H
#define DMSA_ORACLE_H

#include <stdint.h>
#include "../include/dmsa_hip.h" /* POD problem/settings/report structs only */
#include "../include/dmsa_static_points.h"
#include "../include/dmsa_window_setup.h" /* dmsa_traj_state POD only */
#include "../include/dmsa_wire_formats.h" /* dmsa_pointcloud2 POD + sensor enum only */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- rotation / interpolation helpers (helpers.h:24-65, ConsecutivePoses.h:26-67) ---- */
void orc_axang2rotm(const double* axang3, double* R9_colmajor);
void orc_rotm2axang(const double* R9_colmajor, double* axang3);
void orc_slerp(const double* aa1, const double* aa2, double t, double* out3);
void orc_relative2global(int n, const double* rel_o, const double* rel_t, double* glob_o, double* glob_t);
void orc_global2relative(int n, const double* glob_o, const double* glob_t, double* rel_o, double* rel_t);
/* boost::math::barycentric_rational<double>(x, y, n, d)(t) for nt query points */
int  orc_barycentric_rational(const double* x, const double* y, int n, int d, const double* t, int nt, double* out);

/* ---- dense pose tables -------------------------------------------------------------- */
/* updateTrajDenseTforms (ContinuousTrajectory.h:189-226) for the relative poses in p.
 * table: n_total x 12 floats ([R|t] row-major 3x4); dense_transl (optional): 3 x n_total col-major doubles */
int orc_window_pose_table(const dmsa_window_problem* p, float* table, double* dense_transl);
/* include/dmsa_detmath.h on the host: fn 0 sin, 1 cos, 2 acos, 3 atan2(y, x) */
int orc_detmath_eval(int fn, const double* x, const double* y, long n, double* out);
/* per-keyframe transforms of MapManagement::updateGlobalPoints (MapManagement.h:128-138): F x 12 floats */
int orc_keyframe_pose_table(const dmsa_keyframe_problem* p, float* table);
/* p_g = T[row] * (x,y,z,1) exactly as Matrix4f*Vector4f evaluates (ContinuousTrajectory.h:151) */
void orc_transform_points(const float* table, const float* xyz4, const int32_t* row, int64_t n, float* out4);

/* ---- PCL-exact voxelisation (OctreePointCloud; DmsaOptimizer.h:282-298) ---------------- */
int orc_voxelize(const float* xyz4, int64_t n, double resolution, dmsa_voxel_level_info* info,
                 uint64_t* leaf_code /* n */, uint32_t* key_xyz /* n x 3 */, int32_t* sorted_point_idx /* n */);

/* ---- Gaussians: reset + createGaussianSets x2 + updateRebalancingWeights --------------- */
typedef struct orc_gaussians orc_gaussians;
orc_gaussians* orc_build_gaussians(const float* xyz4, const float* normal4 /* or NULL */, const int32_t* ids, int64_t n,
                                   float min_grid_size, const dmsa_settings* s);
void    orc_gaussians_free(orc_gaussians* g);
int32_t orc_gaussians_count(const orc_gaussians* g);
int32_t orc_gaussians_count_level1(const orc_gaussians* g);
int64_t orc_gaussians_memberships(const orc_gaussians* g);
void    orc_gaussians_get(const orc_gaussians* g, int32_t* seg_offset, int32_t* member_idx, float* info_mats, float* weights);
/* intermediate results of the fit: subset.colwise().mean() (M x 3), the covariance before limitCovariance (M x 9, column-major) and
 * pow(-1) of the member counts before the division by their mean (M) -- the 'fit_sums' stage of tests/ref_stage_checks.py */
void    orc_gaussians_get_fit(const orc_gaussians* g, float* mean3, float* cov9, float* raw_weights);
/* overwrite information matrices / weights (stage-level parity: feed the HIP path's Gaussians to the oracle) */
void    orc_gaussians_set_info(orc_gaussians* g, const float* info_mats, const float* weights);
/* updateErrorTerms rows 0..M-1 (DmsaOptimizer.h:242-268) on the given global points */
/* DenseBase::mean() of n contiguous floats whose first entry lies offset_floats behind a 16-byte boundary (Eigen 3.4 linear redux, SSE2) */
float   orc_eigen_mean_f32(const float* x, int64_t n, int64_t offset_floats);
void    orc_eval_residuals(const orc_gaussians* g, const float* xyz4_global, double* e_out);

/* ---- whole optimizeSet ------------------------------------------------------------------- */
typedef struct orc_iter_trace {
    int32_t M, M1;
    int64_t Mm;
    double  error0;
    double  step_norm;
    int32_t best_k;
    int32_t pad;
} orc_iter_trace;
/* global_out (optional): (N+S) x 4 floats after the final updateGlobalPoints; trace (optional): capacity entries.
 * fixed_iters != 0 disables the no-improvement / epsilon exits (benchmarking, mirrors DMSA_FLAG_FIXED_ITERS). */
/* evaluation-parallel variant of the CPU baseline (OpenMP over the P forward differences / 9 line-search trials; without IMU rows
 * results are bit-identical to 1 thread).  Default 1 = the reference's execution order. */
void orc_set_threads(int n);
int orc_get_threads(void);
/* L1 data cache size of the machine the reference runs on: Eigen derives the depth blocking kc of centered^T * centered from it
 * (Gaussians.h:147; 32768 -> kc <= 680, 49152 -> 1016).  Default 32768. */
void orc_set_eigen_l1_bytes(int bytes);
int orc_get_eigen_l1_bytes(void);
int64_t orc_eigen_gemm_kc(int64_t depth);
/* one coefficient of a (1 x n) * (n x 1) slice of that product: sum_k a[k] * b[k] in Eigen 3.4's float order */
float orc_eigen_gemm_dot_f32(const float* a, const float* b, int64_t n);
/* EigenSolver<Matrix3f> of Eigen 3.4.0 as restated in oracle/eigensolver3f.h (Gaussians.h:184-188), limitCovariance (:181-201), telemetry */
void orc_eigensolver3f(const float* A9_colmajor, int64_t count, float* evals_re, float* evals_im, float* V9_colmajor, int32_t* iterations, int32_t* info,
                       int32_t* pairs);
void orc_limit_covariance(const float* cov9_colmajor, int64_t count, float* out9_colmajor);
void orc_limitcov_stats(int64_t* out5, int reset);
void orc_info_from_covariance(const float* cov9_colmajor, int64_t count, float* info9_colmajor);
void orc_limitcov_eigenpairs(const float* cov9_colmajor, int64_t count, float* evals3, float* V9_colmajor);
int orc_optimize_window(dmsa_window_problem* p, const dmsa_settings* s, dmsa_report* rep, float* global_out,
                        orc_iter_trace* trace, int32_t trace_capacity, int32_t fixed_iters);
int orc_optimize_keyframes(dmsa_keyframe_problem* p, const dmsa_settings* s, dmsa_report* rep, float* global_out,
                           orc_iter_trace* trace, int32_t trace_capacity, int32_t fixed_iters);

/* additional error rows for given relative poses (updateImuError / Gravity / Odometry); returns row count */
int orc_window_additional_errors(const dmsa_window_problem* p, double* rows_out);
int orc_keyframe_additional_errors(const dmsa_keyframe_problem* p, double* rows_out);

/* one numeric-Jacobian + LM step on given residual batches (DmsaOptimizer.h:107-113); for stage parity */
/* iteration 0 of optimizeSet stage by stage into the 'DMSAST03' file of dmsa_lidar_slam_amd/dump.py (what oracle/ref_harness/ref_main.cpp
   writes from the real reference); inject_info / inject_weights (or NULL, with inject_M = the expected number of Gaussians) replace the
   fitted information matrices / weights before the residuals are evaluated */
int orc_stage_dump_window(dmsa_window_problem* p, const dmsa_settings* s, const float* inject_info, const float* inject_weights, int32_t inject_M, const char* path);
int orc_stage_dump_keyframes(dmsa_keyframe_problem* p, const dmsa_settings* s, const float* inject_info, const float* inject_weights, int32_t inject_M, const char* path);
int orc_lm_step_from_jacobian(const double* e0, const double* J /* col-major rows x P */, int32_t rows, int32_t P, double lambda, double alpha, double* H_out,
                              double* g_out, double* step_out);
int orc_lm_step(const double* e0, const double* e_batch /* P x rows */, int32_t rows, int32_t P, double h, double lambda,
                double alpha, double* H_out /* PxP col-major */, double* g_out, double* step_out);

/* ---- addStaticPoints selection / getOverlap (DmsaSlam.h:264-414), randomGridDownsampling (helpers.h:67-182) */
int orc_radius_exists(const float* cloud, int64_t n_cloud, const float* query, int64_t n_query, float radius, uint8_t* flag_out, int brute);
int orc_select_static_points(const dmsa_static_select_problem* p, float* static_xyz_out, int32_t* static_id_out, int64_t capacity,
                             int32_t* overlap_per_keyframe, dmsa_static_select_result* res);
int orc_get_overlap(const float* pc1, int64_t n1, const float* pc2, int64_t n2, float maxDistOverlap, float* overlap_out, int64_t* num_corresp_out);
void orc_glibc_rand(uint32_t seed, int32_t count, int32_t* out); /* the first `count` values of rand() after srand(seed) */
int orc_random_grid_downsampling(const float* xyz, int64_t n, float grid_size, uint32_t seed, int32_t* picked_index_out, int64_t capacity,
                                 int64_t* num_out);
/* DmsaSlam::preProcess (DmsaSlam.h:569-634); tform = Eigen storage (column-major) of lidarToImuTform */
int orc_preprocess_scan(const float* raw_xyz, int64_t n, int32_t max_num_points_per_scan, float min_dist_ds, float min_dist, uint32_t seed,
                        const float* tform, float* xyz_out, int32_t* src_index_out, int64_t capacity, int64_t* num_out, float* grid_size_out);

/* ---- ImuBuffer (ImuBuffer.h:14-175), ImuPreintegration (ImuPreintegration.h:23-139) and the setup half of
 * ContinuousTrajectory (ContinuousTrajectory.h:228-568); same array conventions as include/dmsa_window_setup.h */
void* orc_imu_buffer_create(int32_t max_num_meas);
void orc_imu_buffer_destroy(void* b);
void orc_imu_buffer_add(void* b, const double* acc, const double* ang_vel, double stamp);
int orc_imu_buffer_closest(const void* b, double t, double* acc_out, double* ang_vel_out, double* timediff_out);
void orc_imu_buffer_state(const void* b, int32_t* num_updates, int32_t* oldest_index, double* bias_gyr);
int orc_traj_dims(double t_min, double t_max, double dt_res, double* horizon_out, int32_t* n_total_out);
int orc_traj_grids(double horizon, double dt_res, int32_t n_total, int32_t C, double* traj_time_out, double* stamps_out, int32_t* param_indices_out);
int orc_traj_tform_indices(const double* point_stamps, int64_t n, double t0, const double* traj_time, int32_t n_total, int32_t* out);
int orc_traj_transfer_imu(const void* b, double t0, const double* traj_time, int32_t n_total, double* acc_meas_out, double* ang_vel_meas_out, double* worst);
int orc_traj_preint_factors(int32_t n_total, int32_t C, const int32_t* paramIndices, double dt_res, const double* accMeas, const double* angVelMeas,
                            const double* gyr_cov, const double* acc_cov, double* preintImuRots, double* preintRelPositions, double* preintRelVelocity,
                            double* CovPVRot_inv, double* preintPosComplHor);
int orc_traj_update_initial_guess(int32_t* is_initialized, dmsa_traj_state* cur, dmsa_traj_state* old_traj, int32_t use_imu);
int orc_traj_submap_gravity_estimate(const dmsa_traj_state* s, const double* preintPosComplHor, double* gravity_imu);

/* ---- wire formats (src/dmsa_slam_ros.cpp:374-486, OutputManagement.h:80-182) */
int orc_decode_pointcloud2(const dmsa_pointcloud2* msg, int32_t sensor, float* xyz_out, double* stamp_out, int32_t* id_out);
int orc_format_tum_pose(double stamp, const double* pos, const double* orient, char* out, int32_t cap);
int orc_compose_nonkeyframe_pose(const double* keyframePos, const double* keyframeOrient, const double* Translation, const double* Orientation, double* globalPos,
                                 double* globalOrient);

/* ---- keyframe creation (DmsaSlam.h:469-567); exhaustive neighbour search + pcl::NormalEstimation restated */
int orc_update_normals(const float* xyz, int64_t n, int32_t k, const float* viewpoint, float* normal_out, int32_t* nn_index_out);
int orc_make_keyframe_cloud(const float* global_xyz, const int32_t* ids, int64_t n, float min_grid_size, uint32_t seed, const double* pos0, const double* orient0,
                            float* xyz_local_out, float* normal_out, int32_t* ring_out, int32_t* src_index_out, int64_t capacity, int64_t* num_out);

#ifdef __cplusplus
}
#endif
#endif
