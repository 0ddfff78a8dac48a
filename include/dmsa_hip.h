/*
 * dmsa_hip.h — C ABI of the MI355X-native DMSA inner loop.
 *
 * Drop-in boundary for ONE path of davidskdds/DMSA_LiDAR_SLAM: the body of
 *   DmsaOptimizer<PointT>::optimizeSet            (include/DMSA/DmsaOptimizer.h:54-150)
 * driven on the two concrete problem models
 *   ContinuousTrajectory : OptimizablePointSet<PointStampId>   (ContinuousTrajectory.h:24-669)
 *   MapManagement        : OptimizablePointSet<PointNormal>    (MapManagement.h:20-390)
 *
 * The reference has no FFI layer; its seam is a C++ template + nine virtuals
 * (OptimizablePointSet.h:18-56).  A GPU library cannot call per-point virtuals
 * back, so this ABI absorbs the two concrete models: the caller hands over the
 * state those virtuals read (plain pointers, Eigen column-major layouts so real
 * Eigen objects can be passed with .data()), the library owns all device memory.
 *
 * Conventions
 *   - every entry point returns 0 (DMSA_OK) or a negative dmsa_status; nothing throws
 *     across the ABI; no global state; one context per host thread / GPU.
 *   - "3xn col-major double" == Eigen::Matrix3Xd::data().
 *   - point arrays are float[n][4] == pcl PointXYZ-style data[4] (PointStampId.h:33-45),
 *     the 4th float is ignored on input (the reference keeps it at 1.0f).
 *   - the library FAILS (DMSA_ERR_NO_DEVICE) when no HIP device is usable; there is no
 *     CPU fallback behind this ABI.
 */
#ifndef DMSA_HIP_H
#define DMSA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmsa_ctx dmsa_ctx;

typedef enum dmsa_status {
    DMSA_OK = 0,
    DMSA_ERR_INVALID = -1,     /* bad argument / problem not uploaded                      */
    DMSA_ERR_NO_DEVICE = -2,   /* no usable HIP device — never falls back to the CPU       */
    DMSA_ERR_HIP = -3,         /* a HIP runtime call failed (dmsa_last_error has the text) */
    DMSA_ERR_DEPTH = -4,       /* voxel lattice deeper than 21 levels (> 63-bit leaf code) */
    DMSA_ERR_NOMEM = -5
} dmsa_status;

/* Why optimizeSet left its loop (DmsaOptimizer.h:89-93, :116-122, :130-134, :139-143). */
typedef enum dmsa_stop_reason {
    DMSA_STOP_NUM_ITER = 0,        /* ran settings.num_iter iterations                     */
    DMSA_STOP_FEW_GAUSSIANS = 1,   /* numPointSets < min_num_gaussians                     */
    DMSA_STOP_NAN = 2,             /* NaN in the LM step, parameters restored              */
    DMSA_STOP_NO_IMPROVEMENT = 3,  /* adaptiveStepSize returned 0                          */
    DMSA_STOP_EPSILON = 4          /* ||step|| < epsilon                                   */
} dmsa_stop_reason;

/* == DmsaOptimSettings (DmsaOptimizer.h:25-39), same defaults via dmsa_default_settings. */
typedef struct dmsa_settings {
    int32_t num_iter;                /* 15     */
    double  epsilon;                 /* 1e-5   */
    int32_t use_analytic_jacobi;     /* false — never read by the reference either (:29)  */
    double  step_length_optim;       /* 0.05   */
    double  max_step;                /* 0.01   */
    int32_t gauss_split;             /* false  */
    float   grid_size_1_factor;      /* 2.0    */
    float   grid_size_2_factor;      /* 5.0    */
    int32_t min_num_points_per_set;  /* 6      */
    int32_t min_num_gaussians;       /* 30     */
    float   lambda_diag;             /* 1e-5   */
    int32_t use_centralization;      /* true   */
} dmsa_settings;

/* State of a ContinuousTrajectory that the hot path reads/writes.
 * Replaces: get/setPoseParameters (ContinuousTrajectory.h:119-127), updateGlobalPoints (:129-156),
 * updateTrajDenseTforms (:189-226), updateAdditionalErrors/updateImuError (:107-117, :603-663),
 * centralize/decentralize (:75-100), getIdOfPoint (:665-668). */
typedef struct dmsa_window_problem {
    int32_t        num_control_poses;   /* C = controlPoses.numPoses                              */
    double*        rel_orient;          /* 3xC col-major, controlPoses.relativePoses.Orientations — IN/OUT */
    double*        rel_transl;          /* 3xC col-major, controlPoses.relativePoses.Translations — IN/OUT */
    const double*  stamps;              /* C, controlPoses.stamps (strictly increasing)           */
    int32_t        n_total;             /* dense pose count                                       */
    const double*  traj_time;           /* n_total, trajTime                                      */
    int64_t        num_points;          /* N window points (all clouds of regPcBuffer, chronological) */
    const float*   xyz_local;           /* N x 4, local (sensor-frame) coordinates                */
    const int32_t* tform_idx;           /* N, tformIdPerPoint flattened (registerPcBuffer :240-260) */
    const int32_t* ring_id;             /* N, PointStampId::id                                    */
    int64_t        num_static;          /* S static map points appended by addStaticPoints (:158-172) */
    const float*   xyz_static;          /* S x 4, world frame, NOT yet centralised                */
    const int32_t* ring_id_static;      /* S                                                      */
    float          min_grid_size;       /* OptimizablePointSet::minGridSize                       */
    /* IMU factor rows (updateImuError); ignored when use_imu == 0 */
    int32_t        use_imu;             /* useImuErrorTerms                                       */
    double         dt_res;
    double         balancing_imu;
    double         gravity[3];
    const int32_t* param_indices;       /* C, paramIndices                                        */
    const double*  preint_rot;          /* C x 9, preintImuRots[k] col-major (entry 0 unused)     */
    const double*  preint_pos;          /* C x 3, preintRelPositions                              */
    const double*  preint_vel;          /* C x 3, preintRelVelocity                               */
    const double*  cov_pvrot_inv;       /* C x 81, CovPVRot_inv[k] col-major                      */
} dmsa_window_problem;

/* State of a MapManagement (sub)map that the hot path reads/writes.
 * Replaces: get/setPoseParameters (MapManagement.h:192-202), updateGlobalPoints (:120-149),
 * updateAdditionalErrors / Gravity / Odometry (:162-252), getIdOfPoint (:204-208). */
typedef struct dmsa_keyframe_problem {
    int32_t        num_frames;          /* keyframeDataBuffer.getNumElements()                    */
    double*        rel_orient;          /* 3xF col-major keyframePoses.relativePoses.Orientations — IN/OUT */
    double*        rel_transl;          /* 3xF col-major                                   — IN/OUT */
    const int64_t* frame_offset;        /* F+1 prefix of points per keyframe                      */
    const float*   xyz_local;           /* n x 4 local PointNormal xyz, keyframes concatenated    */
    const float*   normal_local;        /* n x 4 local normals                                    */
    const int32_t* ring_id;             /* n, MapManagement::ringIds                              */
    float          min_grid_size;
    int32_t        use_gravity;         /* useGravityErrorTerms                                   */
    int32_t        use_odometry;        /* useOdometryErrorTerms                                  */
    double         gravity[3];
    double         cov_grav_inv[9];     /* col-major                                              */
    double         balancing_grav;
    double         balancing_odom;
    const double*  measured_gravity;    /* F x 3                                                  */
    const int32_t* gravity_plausible;   /* F                                                      */
    const double*  odom_rel_transl;     /* F x 3, KeyframeData::relativeTransl                    */
    const double*  odom_rel_orient_mat; /* F x 9 col-major, KeyframeData::relativeOrientMat       */
    double         odom_transl_cov_inv[9];
    double         odom_orient_cov_inv[9];
} dmsa_keyframe_problem;

typedef struct dmsa_report {
    int32_t iterations;        /* loop bodies entered                                            */
    int32_t stop_reason;       /* dmsa_stop_reason                                               */
    int32_t num_gaussians;     /* M of the last iteration                                        */
    int32_t num_gaussians_l1;  /* of which from the first resolution                             */
    int64_t num_memberships;   /* Mm = sum of points over the Gaussians of the last iteration    */
    double  error0;            /* e^T e at the start of the last iteration                       */
    double  last_step_norm;
    int32_t last_line_search_k;
    int32_t evaluations;       /* forward evaluations done in total                              */
} dmsa_report;

/* per-iteration record of the last optimize call (diagnostics / parity tests) */
typedef struct dmsa_iter_trace {
    int32_t M, M1;             /* Gaussians in total / from the first resolution                  */
    int64_t Mm;                /* memberships                                                     */
    double  error0;            /* e^T e before the step                                           */
    double  step_norm;         /* norm of the clamped LM step                                     */
    int32_t best_k;            /* adaptiveStepSize result                                         */
    int32_t pad;
} dmsa_iter_trace;

/* context flags.  flags = 0 is the path that meets the 1e-4 m / 1e-4 rad bar: per-Gaussian sums in the reference's member order
 * (DmsaOptimizer.h:247-264), LM step through the explicit inverse (:113), dense pose tables built on the device with the shared
 * operation sequences of dmsa_detmath.h -- poses bit-identical to the CPU restatement of the reference. */
#define DMSA_FLAG_POSE_TABLE_HOST 0x1u /* build the dense pose tables on host threads instead of the device kernel (same
                                          arithmetic, same bits; kept for debugging)                                   */
#define DMSA_FLAG_FIXED_ITERS     0x2u /* benchmarking: ignore the no-improvement / epsilon exits    */
#define DMSA_FLAG_STAGE_TIMERS    0x8u /* also time voxelisation / fit / pose tables / normal equations with HIP events (each
                                          event pair costs ~10 us of GPU idle; the correspondence kernel is always timed)  */
#define DMSA_FLAG_MIRROR_SUMS     0x4u /* round-1 name of what is now the default; accepted and ignored                 */
/* (0x10 was DMSA_FLAG_FAST_SUMS, the wave-parallel sums of rounds 1-4: slower than the reference-order path since round 3 and outside
 * the 1e-4 tolerance; retired in round 5.  dmsa_create returns DMSA_ERR_INVALID for flag bits it does not know.) */

int  dmsa_create(int device, uint32_t flags, dmsa_ctx** out);
void dmsa_destroy(dmsa_ctx* ctx);
const char* dmsa_last_error(const dmsa_ctx* ctx);
void dmsa_default_settings(dmsa_settings* s);

/* ---- whole optimizeSet (the drop-in calls) ---------------------------------------------- */
/* == slidingWindowOptimizer.optimizeSet(*currTraj, optimSettingsSlidingWindow)  DmsaSlam.h:166 */
int dmsa_optimize_window(dmsa_ctx* ctx, dmsa_window_problem* p, const dmsa_settings* s, dmsa_report* rep);
/* == keyframeMapOptimizer.optimizeSet(*currSubmap, optimSettingsMap)            DmsaSlam.h:228 */
int dmsa_optimize_keyframes(dmsa_ctx* ctx, dmsa_keyframe_problem* p, const dmsa_settings* s, dmsa_report* rep);
/* Same loop on the problem already resident in HBM (dmsa_window_upload / dmsa_keyframes_upload or a previous optimize
 * call); poses continue from the context's current state and are returned through dmsa_get_poses. */
int dmsa_optimize_resident(dmsa_ctx* ctx, const dmsa_settings* s, dmsa_report* rep);
/* DmsaOptimizer<PointT>::adaptiveStepSize (DmsaOptimizer.h:152-182; public in the reference, called by optimizeSet only) on the resident problem
 * and its current Gaussians: nine trial evaluations at params + 0.1 k step, k = 1 .. 9; `params` (P doubles, the raw parameters on entry) becomes
 * the trial with the smallest e^T e if that beats error0 (strict '<'), *best_k its k -- 0 if none does.  Like the reference's object, the resident
 * poses are left at the LAST trial (raw + 0.9 step), whatever the outcome.  Needs Gaussians (dmsa_build_gaussians or a previous optimize call). */
int dmsa_adaptive_step_size(dmsa_ctx* ctx, double* params, const double* step, double error0, int32_t* best_k);
/* current relative poses of the resident problem: 3 x n col-major doubles each (n = control poses / keyframes) */
int dmsa_get_poses(dmsa_ctx* ctx, double* rel_orient, double* rel_transl);
/* final globalPoints of the last optimize call (DmsaOptimizer.h:149); n x 4 floats */
int dmsa_get_global_points(dmsa_ctx* ctx, float* xyz_out, int64_t capacity_points);

/* ---- stage-level entry points (used by the parity tests and the benchmark) -------------- */
int dmsa_window_upload(dmsa_ctx* ctx, const dmsa_window_problem* p);       /* points resident in HBM */
int dmsa_keyframes_upload(dmsa_ctx* ctx, const dmsa_keyframe_problem* p);
/* ContinuousTrajectory::centralize()/decentralize() on the uploaded problem (poses + static points) */
int dmsa_centralize(dmsa_ctx* ctx);
int dmsa_decentralize(dmsa_ctx* ctx);
/* current parameter vector (getPoseParameters) of the uploaded problem, P doubles */
int dmsa_get_params(dmsa_ctx* ctx, double* params, int32_t* P);
int dmsa_set_params(dmsa_ctx* ctx, const double* params);
/* == updateAdditionalErrors() + getAdditionalErrorTerms() (OptimizablePointSet.h:27-31; ContinuousTrajectory.h:102-117 IMU rows,
 * MapManagement.h:162-252 gravity then odometry rows) for the CURRENT pose parameters of the uploaded problem (the chain is
 * re-linked first, like updateGlobalPoints does).  rows_out: capacity doubles; num_out = number of rows (0 when the model has none). */
int dmsa_additional_errors(dmsa_ctx* ctx, double* rows_out, int32_t capacity, int32_t* num_out);
/* Build B dense pose tables from B parameter vectors (B x P row-major doubles).
 * tables_out (optional, host): B x n_rows x 12 floats, each row = 3x4 [R|t] row-major.
 * == setPoseParameters + updateTrajDenseTforms (window) / per-keyframe transforms (keyframes). */
int dmsa_pose_tables(dmsa_ctx* ctx, int32_t B, const double* params, float* tables_out);
/* Same, but the caller supplies the tables (B x n_rows x 12 floats). */
int dmsa_set_pose_tables(dmsa_ctx* ctx, int32_t B, const float* tables);
int dmsa_num_table_rows(dmsa_ctx* ctx, int32_t* n_rows);
/* updateGlobalPoints() with pose table b of the current batch; xyz_out optional (n x 4). */
int dmsa_transform_points(dmsa_ctx* ctx, int32_t b, float* xyz_out);
/* currentGauss.reset() + createGaussianSets at both resolutions + updateRebalancingWeights
 * on the current global points (DmsaOptimizer.h:78-96). */
int dmsa_build_gaussians(dmsa_ctx* ctx, const dmsa_settings* s, int32_t* M_out, int64_t* Mm_out);
/* updateErrorTerms for the B tables of the current batch: e_out = B x M doubles (Gaussian rows only,
 * row order = leaf DFS order, first resolution then second). */
int dmsa_eval_residuals(dmsa_ctx* ctx, double* e_out);
/* H = J^T J + lambda I, g = J^T e0 with J.col(k) = (e_{k+1} - e_0)/h from the last batch of 1+P evaluations;
 * extra_rows (optional) = (1+P) x a additional error rows computed by the caller. H: P x P col-major. */
int dmsa_normal_equations(dmsa_ctx* ctx, int32_t P, int32_t a, const double* extra_rows, double h, double lambda,
                          double* H_out, double* g_out);

/* introspection for the parity tests */
typedef struct dmsa_voxel_level_info {
    double  resolution;      /* octree resolution_ (float product widened, DmsaOptimizer.h:82)        */
    double  min_xyz[3];      /* final min_x_/min_y_/min_z_                                             */
    int32_t depth;           /* final octree_depth_                                                    */
    int32_t num_events;      /* bounding-box growth events                                            */
    int64_t num_leaves;      /* occupied leaves                                                       */
    int64_t num_valid;       /* finite points inserted                                                */
} dmsa_voxel_level_info;
int dmsa_get_voxel_level(dmsa_ctx* ctx, int32_t level, dmsa_voxel_level_info* info,
                         uint64_t* leaf_code /* n, per point, DFS code */, uint32_t* key_xyz /* n x 3 */,
                         int32_t* sorted_point_idx /* n, leaf DFS order then ascending index */);
int dmsa_get_gaussians(dmsa_ctx* ctx, int32_t* seg_offset /* M+1 */, int32_t* member_idx /* Mm */,
                       float* info_mats /* M x 9 col-major */, float* weights /* M */);

/* The voxelisation's device primitives on caller data, for tests (host arrays in, host arrays out, synchronous):
 * dmsa_sort_pairs   stable sort of (u32 key, u32 value) pairs on key bits [0, end_bit) with the library's own radix sort
 *                   (csrc/radix_sort.hip) -- the order of pcl::octree's depth-first leaf iteration with ascending point index in a leaf
 *                   is exactly a stable sort by leaf code (octree_pointcloud.hpp, DmsaOptimizer.h:284-316);
 * dmsa_leaf_segments  the segmentation of sorted codes into leaves by the single-pass kernel: leaf_of_pos[i] = 1-based index of the
 *                   leaf position i belongs to, leaf_start[0 .. num_leaves] and the number of leaves; the code 1 << code_bits marks
 *                   non-finite points (sorted to the end) and belongs to no leaf. */
int dmsa_sort_pairs(dmsa_ctx* ctx, const uint32_t* keys, const uint32_t* values, int64_t n, uint32_t end_bit, uint32_t* keys_sorted,
                    uint32_t* values_sorted);
int dmsa_leaf_segments(dmsa_ctx* ctx, const uint32_t* codes_sorted, int64_t n, uint32_t code_bits, int32_t* leaf_of_pos /* n */,
                       int32_t* leaf_start /* n + 1 */, int32_t* num_leaves);

/* Introspection of the default path's correspondence kernels: the double sum of a long Gaussian (DmsaOptimizer.h:259-264) is computed
 * as a parallel reduction whenever integer bounds prove that the reference's member-by-member chain cannot round;
 * this is the number of (Gaussian, evaluation sub-batch) sums for which the proof failed and the chain was run instead, since the
 * last reset.  Results are the same either way; the counter only says how often the slow way was taken.  Synchronises the device. */
int dmsa_serial_fallback_sums(dmsa_ctx* ctx, int32_t reset, uint64_t* count);

/* Host-only: the Levenberg-Marquardt step of DmsaOptimizer.h:110-113, step = (-alpha * (H + lambda I)^-1) * g, with the explicit
 * inverse the reference forms (`H_damped` = H + lambda I, P x P column-major, symmetric or not).  `threads` > 1 spreads the row
 * updates of every pivot step over that many host threads for P >= 64 -- the result does not depend on it (tested bit for bit);
 * inside optimizeSet the context's worker pool plays that part.  No device needed. */
int dmsa_lm_solve(const double* H_damped, const double* g, int32_t P, double alpha, int32_t threads, double* step);

/* The same step computed ON THE DEVICE, as the device-resident loop of optimizeSet does (csrc/loop_kernels.hip): one workgroup for
 * P <= 64, one workgroup per block of 8 columns of [H | I] beyond, panels of 8 pivot columns handed from owner to owner.  Followed by
 * the NaN test and the clamp of DmsaOptimizer.h:116-128 (`max_step` = +inf disables the clamp; *nan_out = 1 and an undefined step when
 * the step holds a NaN).  Bit-identical to dmsa_lm_solve (tests). */
int dmsa_lm_solve_device(dmsa_ctx* ctx, const double* H_damped, const double* g, int32_t P, double alpha, double max_step, double* step,
                         int32_t* nan_out);

/* Evaluates include/dmsa_detmath.h ON THE DEVICE: fn 0 sin(x), 1 cos(x), 2 acos(x), 3 atan2(y, x); n doubles each (y may be NULL for
 * fn < 3).  The pose-table kernels (ContinuousTrajectory.h:189-226, MapManagement.h:140-147) take their trigonometry from that
 * header; the tests compare these device results bit for bit with the same header compiled for the host. */
int dmsa_detmath_eval(dmsa_ctx* ctx, int32_t fn, const double* x, const double* y, int64_t n, double* out);

/* timing of the last optimize / stage calls, milliseconds measured with HIP events on the library stream */
typedef struct dmsa_timing {
    double residual_kernel_ms;   /* accumulated time inside the correspondence kernel                 */
    int64_t residual_launches;
    int64_t residual_evaluations;
    double residual_algorithmic_bytes; /* sum over launches of bytes(B) = 16*Mm + 48*M + B*(48*rows + 8*M):
                                          what a launch of B evaluations must move when it reads the members once    */
    double residual_unit_bytes;        /* sum over launches of B * bytes(1), bytes(1) = 16*Mm + 56*M + 48*rows: the per-unit
                                          (per-evaluation) figure times the units each launch processed                */
    double voxelize_ms;
    double gaussian_fit_ms;
    double pose_table_ms;
    double normal_eq_ms;
    double total_ms;
} dmsa_timing;
int dmsa_get_timing(dmsa_ctx* ctx, dmsa_timing* t, int32_t reset);
int dmsa_synchronize(dmsa_ctx* ctx);
/* copies up to `capacity` per-iteration records of the last optimize call; returns the number written (>= 0) */
int dmsa_get_trace(dmsa_ctx* ctx, dmsa_iter_trace* out, int32_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_HIP_H */
