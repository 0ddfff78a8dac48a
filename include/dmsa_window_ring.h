/*
 * dmsa_window_ring.h — the scans of the sliding window stay in HBM from one optimizeSet to the next.
 *
 * DmsaSlam::processPointCloud (DmsaSlam.h:116-204) keeps the last `config.n_clouds` scans (Config.h:19, DmsaSlam.h:64, :140) in a ring buffer (RingBuffer.h:31-88,
 * PointCloudBuffer.h:24-49): every new scan replaces the oldest one, the other scans of the window are the ones the previous
 * optimizeSet already saw.  dmsa_optimize_window (dmsa_hip.h) takes the whole window as host arrays on every call -- 30 MB over PCIe
 * for a 10 x 131 072-point window.  With this header a caller uploads ONE scan per window:
 *
 *   dmsa_window_ring_create   once: the ring and every device buffer of the hot path, sized for the full window (no hipMalloc later)
 *   dmsa_window_ring_push     per scan: == pcBuffer->addElem(scan) (RingBuffer.h:67-88)
 *   dmsa_window_upload_from_ring   per window: == prepareTrajectoryForOptimization's registerPcBuffer (ContinuousTrajectory.h:228-261)
 *                             on the resident scans + the control poses / static points of this window
 *   dmsa_optimize_resident, dmsa_get_poses (dmsa_hip.h)
 *
 * The window's points are the ring's scans oldest first (RingBuffer::at is chronological), the pose-table row of a point is
 * min(lower_bound(trajTime, stamp - t0), n_total - 1) computed on the device from the resident stamps (:240-260), so the resident
 * problem is bit for bit the one dmsa_window_upload builds from host arrays (tests/test_gpu_window_ring.py).
 */
#ifndef DMSA_WINDOW_RING_H
#define DMSA_WINDOW_RING_H

#include "dmsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmsa_window_ring_config {
    int32_t num_scans;            /* scans per window: config.n_clouds (RingBuffer capacity)                  */
    int64_t max_points_per_scan;  /* capacity of a ring slot                                                     */
    int64_t max_static_points;    /* capacity reserved for the static map points addStaticPoints appends        */
    int32_t max_n_total;          /* dense poses the pose tables are sized for (horizon / dt_res + 1)            */
    int32_t max_control_poses;    /* control poses the evaluation batches are sized for (<= 64)                  */
} dmsa_window_ring_config;

int dmsa_window_ring_create(dmsa_ctx* ctx, const dmsa_window_ring_config* cfg);
/* The newest scan replaces the oldest (or fills the next free slot): xyz_local n x 4 floats (sensor frame), stamps n doubles
 * (PointStampId::stamp), ring_id n.  One host-to-device copy of n x 28 bytes through pinned staging. */
int dmsa_window_ring_push(dmsa_ctx* ctx, const float* xyz_local, const double* stamps, const int32_t* ring_id, int64_t n);
int dmsa_window_ring_points(dmsa_ctx* ctx, int32_t* scans_out, int64_t* points_out);
/* The resident scans become the window problem: `p` carries control poses, time grid, static points, IMU factors as for
 * dmsa_window_upload; p->xyz_local / tform_idx / ring_id are ignored (may be NULL) and p->num_points must be 0 or the ring's point count.
 * t0: the stamp trajTime[0] refers to (registerPcBuffer subtracts it from every point stamp). */
int dmsa_window_upload_from_ring(dmsa_ctx* ctx, const dmsa_window_problem* p, double t0);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_WINDOW_RING_H */
