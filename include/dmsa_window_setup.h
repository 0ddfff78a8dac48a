/*
 * dmsa_window_setup.h — C ABI of the producers of the hot path's inputs: everything
 * DmsaSlam::prepareTrajectoryForOptimization (include/DMSA/DmsaSlam.h:416-461) asks of ContinuousTrajectory before
 * optimizeSet runs, and the IMU ring buffer it reads.
 *
 *   ContinuousTrajectory::initTraj                include/DMSA/ContinuousTrajectory.h:301-346
 *   ContinuousTrajectory::registerPcBuffer        include/DMSA/ContinuousTrajectory.h:228-261   (tformIdPerPoint)
 *   ContinuousTrajectory::transferImuMeasurements include/DMSA/ContinuousTrajectory.h:348-364
 *   ContinuousTrajectory::updatePreintFactors     include/DMSA/ContinuousTrajectory.h:518-568
 *   ContinuousTrajectory::updateInitialGuess      include/DMSA/ContinuousTrajectory.h:366-468
 *   ContinuousTrajectory::initGravityDir          include/DMSA/ContinuousTrajectory.h:263-299
 *   ContinuousTrajectory::getImuIntegratedParams  include/DMSA/ContinuousTrajectory.h:470-516
 *   ContinuousTrajectory::getSubmapGravityEstimate include/DMSA/ContinuousTrajectory.h:593-601   (measuredGravity of a new keyframe)
 *   ImuPreintegration                             include/DMSA/ImuPreintegration.h:23-139
 *   ImuBuffer                                     include/DMSA/ImuBuffer.h:14-175
 *
 * The outputs are exactly the setup fields of the window problem struct of dmsa_hip.h: stamps, n_total, traj_time, tform_idx,
 * param_indices, preint_rot / preint_pos / preint_vel / cov_pvrot_inv and the initial rel_orient / rel_transl.
 *
 * All of it is O(#poses) double arithmetic on the host, like the reference — except the per-point lower_bound of
 * registerPcBuffer (one binary search per window point), which runs on the device.  Matrices are in Eigen's storage order
 * (column-major); 3 x n arrays hold column k at [3k .. 3k+2].  Same status codes as dmsa_hip.h.
 */
#ifndef DMSA_WINDOW_SETUP_H
#define DMSA_WINDOW_SETUP_H

#include "dmsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ImuBuffer (ImuBuffer.h:14-175): ring buffer of maxNumMeas measurements, gyro bias from the first 50 --------------------- */
typedef struct dmsa_imu_buffer dmsa_imu_buffer;

int  dmsa_imu_buffer_create(int32_t max_num_meas /* ImuBuffer.h:24: 10000 */, dmsa_imu_buffer** out);
void dmsa_imu_buffer_destroy(dmsa_imu_buffer* b);
/* addMeasurement (:46-66): stores acc, ang_vel - bias_gyr, stamp; after exactly 50 updates bias_gyr = mean of the stored angular
 * velocities (the first 50 are stored without bias). */
int  dmsa_imu_buffer_add(dmsa_imu_buffer* b, const double acc[3], const double ang_vel[3], double stamp);
/* getClosestMeasurement (:68-125): the first stored stamp that is not less than t inside the search range(s) the reference uses
 * (NOT the nearest one), its measurement and the time difference with the reference's sign conventions.  DMSA_ERR_INVALID on an
 * empty buffer (the reference would search an inverted range). */
int  dmsa_imu_buffer_closest(const dmsa_imu_buffer* b, double t, double acc_out[3], double ang_vel_out[3], double* timediff_out);
/* bookkeeping the node reads: numUpdates, oldestIndex, bias_gyr, getLatestStamp (:127-135), getOldestStamp (:137-145) */
int  dmsa_imu_buffer_state(const dmsa_imu_buffer* b, int32_t* num_updates, int32_t* oldest_index, double bias_gyr[3], double* latest_stamp,
                           double* oldest_stamp);

/* ---- initTraj (:301-346) -------------------------------------------------------------------------------------------------- */
/* horizon = t_max - t_min + dt_res; n_total = round(horizon / dt_res) + 1 (:307-309) */
int dmsa_traj_dims(double t_min, double t_max, double dt_res, double* horizon_out, int32_t* n_total_out);
/* trajTime = LinSpaced(n_total, 0, horizon) (:322), controlPoses.stamps = LinSpaced(C, 0, horizon) (:331),
 * paramIndices = round(stamps / dt_res) (:334-335) */
int dmsa_traj_grids(double horizon, double dt_res, int32_t n_total, int32_t num_control_poses, double* traj_time_out, double* stamps_out,
                    int32_t* param_indices_out);

/* ---- registerPcBuffer (:240-260): tformIdPerPoint[k] = min(lower_bound(trajTime, stamp_k - t0), n_total - 1) ------------------ */
/* point_stamps: N absolute stamps (PointStampId::stamp) of the window clouds in buffer order.  Runs on the device. */
int dmsa_traj_tform_indices(dmsa_ctx* ctx, const double* point_stamps, int64_t n, double t0, const double* traj_time, int32_t n_total,
                            int32_t* tform_idx_out);

/* ---- transferImuMeasurements (:348-364): accMeas / angVelMeas column k = closest measurement at t0 + trajTime(k) ------------- */
/* worst_timediff_out (optional): largest |timediff|, the quantity behind the reference's warning (> 0.1 s) */
int dmsa_traj_transfer_imu(const dmsa_imu_buffer* b, double t0, const double* traj_time, int32_t n_total, double* acc_meas_out /* 3 x n_total */,
                           double* ang_vel_meas_out /* 3 x n_total */, double* worst_timediff_out);

/* ---- updatePreintFactors (:518-568) with ImuPreintegration (ImuPreintegration.h:53-107) -------------------------------------- */
/* preint_rot C x 9 (col-major 3x3 each, entry 0 = identity), preint_pos / preint_vel C x 3 (entry 0 = 0), cov_pvrot_inv C x 81
 * (entry 0 untouched by the reference: written as zeros here), preint_pos_horizon = preintPosComplHor */
int dmsa_traj_preint_factors(int32_t n_total, int32_t num_control_poses, const int32_t* param_indices, double dt_res, const double* acc_meas,
                             const double* ang_vel_meas, const double gyr_cov[9], const double acc_cov[9], double* preint_rot_out,
                             double* preint_pos_out, double* preint_vel_out, double* cov_pvrot_inv_out, double preint_pos_horizon_out[3]);

/* ---- updateInitialGuess (:366-468) ------------------------------------------------------------------------------------------ */
typedef struct dmsa_traj_state {
    double        t0;                 /* absolute start of the window                                                         */
    double        horizon;
    double        dt_res;
    int32_t       n_total;
    int32_t       num_control_poses;  /* C                                                                                    */
    const double* stamps;             /* C                                                                                    */
    const double* traj_time;          /* n_total                                                                              */
    const double* acc_meas;           /* 3 x n_total (only read when use_imu)                                                 */
    const double* ang_vel_meas;       /* 3 x n_total                                                                          */
    double        gravity[3];         /* (0, 0, -9.805), :345                                                                 */
    double*       rel_orient;         /* 3 x C controlPoses.relativePoses.Orientations  IN/OUT                                */
    double*       rel_transl;         /* 3 x C                                          IN/OUT                                */
    double*       glob_orient;        /* 3 x C controlPoses.globalPoses.Orientations    IN/OUT                                */
    double*       glob_transl;        /* 3 x C                                          IN/OUT                                */
} dmsa_traj_state;

/* cur: the freshly initialised trajectory (poses all zero, what a fresh heap gives the reference's unset matrices); old: the
 * previous window after its optimisation.  First call (*is_initialized == 0): initGravityDir when use_imu, sets the flag, returns
 * (:370-379).  Later calls: interpolate the known part from `old` (slerp / barycentric rational of order 2), then predict the rest
 * by IMU integration (use_imu) or constant relative motion.  old's global poses are refreshed by relative2global (:382). */
int dmsa_traj_update_initial_guess(int32_t* is_initialized, dmsa_traj_state* cur, dmsa_traj_state* old_traj, int32_t use_imu);

/* ---- getSubmapGravityEstimate (:593-601), called by initializeMap / addNewKeyframeToMap (DmsaSlam.h:488-489, :533-534) -------------
 * gravity_imu = (R(first orientation)^T * (last translation - first translation - v_start_w * horizon) - preintPosComplHor)
 *               / (0.5 * horizon^2),   v_start_w = (denseTranslation(1) - denseTranslation(0)) / dt_res.
 * Reads the GLOBAL control poses of `s` (glob_orient / glob_transl), stamps, traj_time, horizon, dt_res. */
int dmsa_traj_submap_gravity_estimate(const dmsa_traj_state* s, const double preint_pos_horizon[3], double gravity_imu_out[3]);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_WINDOW_SETUP_H */
