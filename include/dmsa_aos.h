/*
 * dmsa_aos.h — the reference's own point containers at the C ABI.
 *
 * The reference keeps its clouds as arrays of structures: pcl::PointCloud<PointStampId> (32 bytes per point: float data[4], double
 * stamp, int id, int isStatic -- include/DMSA/PointStampId.h:33-45) for the sliding window, pcl::PointCloud<pcl::PointNormal> (48 bytes:
 * float data[4], float data_n[4], float curvature + padding) per keyframe (KeyframeData.h:20).  dmsa_hip.h takes flat N x 4 float
 * arrays, which forced a caller to repack point by point into std::vectors (31 ms for the bench window, more than the ten iterations
 * it feeds).  Here a cloud is handed over as it lies in memory -- cloud.points.data(), sizeof(PointT), the byte offsets of the
 * fields.  The library's worker threads move what the device needs into pinned staging (window clouds: the 16 bytes x, y, z, id of every
 * 32-byte point, so the scan crosses PCIe as 20 bytes per point like the flat arrays; keyframe clouds: the points as they are), one DMA
 * per cloud overlaps the staging of the next, and a kernel per cloud writes the layout the hot path reads (pose-table row from
 * tformIdPerPoint, range-checked on the device).  Results are bit-identical to the flat entry points fed with the same numbers
 * (tests/test_gpu_aos.py, examples/aos_call_demo.cpp).
 */
#ifndef DMSA_AOS_H
#define DMSA_AOS_H

#include "dmsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* one strided cloud: point i lives at (const char*)base + i * stride */
typedef struct dmsa_aos_view {
    const void*    base;        /* cloud.points.data()                                                                         */
    int64_t        count;       /* cloud.points.size()                                                                         */
    int32_t        stride;      /* sizeof(PointT): 32 for PointStampId, 48 for pcl::PointNormal; a multiple of 4               */
    int32_t        xyz_offset;  /* offsetof(PointT, x): three consecutive floats (0 for every PCL point type)                  */
    int32_t        aux_offset;  /* window clouds / static points: offsetof(PointStampId, id) = 24, an int32 (the ring id, getIdOfPoint,
                                   ContinuousTrajectory.h:665-668); keyframe clouds: offsetof(PointNormal, normal_x) = 16, four floats  */
    const int32_t* index;       /* window clouds: tformIdPerPoint[pc].data() (ContinuousTrajectory.h:240-260), `count` entries;
                                   keyframe clouds: the frame's slice of MapManagement::ringIds; static points: NULL             */
} dmsa_aos_view;

/* dmsa_window_upload (dmsa_hip.h) with the points taken from the scan ring buffer as it is: `clouds` = regPcBuffer->at(0 .. num_clouds-1)
 * in chronological order (RingBuffer.h:31-65, the order of updateGlobalPoints, ContinuousTrajectory.h:137-155), `static_points` = the tail
 * addStaticPoints appended to globalPoints (:158-172; NULL or count 0: none).  The point arrays of `p` (xyz_local, tform_idx, ring_id,
 * xyz_static, ring_id_static, num_points, num_static) are ignored; everything else in `p` is read as by dmsa_window_upload. */
int dmsa_window_upload_aos(dmsa_ctx* ctx, const dmsa_window_problem* p, const dmsa_aos_view* clouds, int32_t num_clouds, const dmsa_aos_view* static_points);
/* dmsa_keyframes_upload with one view per keyframe (keyframeDataBuffer.at(k).pointCloudLocal->points, MapManagement.h:120-149); frame_offset,
 * xyz_local, normal_local and ring_id of `p` are ignored. */
int dmsa_keyframes_upload_aos(dmsa_ctx* ctx, const dmsa_keyframe_problem* p, const dmsa_aos_view* frames, int32_t num_frames);
/* the whole drop-in calls: upload + optimizeSet (DmsaOptimizer.h:54-150) + poses written back into p->rel_orient / rel_transl */
int dmsa_optimize_window_aos(dmsa_ctx* ctx, dmsa_window_problem* p, const dmsa_aos_view* clouds, int32_t num_clouds, const dmsa_aos_view* static_points,
                             const dmsa_settings* s, dmsa_report* rep);
int dmsa_optimize_keyframes_aos(dmsa_ctx* ctx, dmsa_keyframe_problem* p, const dmsa_aos_view* frames, int32_t num_frames, const dmsa_settings* s, dmsa_report* rep);
/* the final updateGlobalPoints (DmsaOptimizer.h:149) into the caller's own globalPoints: x, y, z of point i are written at
 * base + i * stride + xyz_offset (and, normal_offset >= 0, the rotated normal at + normal_offset); every other byte of the points is
 * left alone. */
int dmsa_get_global_points_aos(dmsa_ctx* ctx, void* base, int64_t count, int32_t stride, int32_t xyz_offset, int32_t normal_offset);
/* The resident scan ring (dmsa_window_ring.h: one scan uploaded per window instead of the whole window) fed from PCL containers:
 * dmsa_window_ring_push with the scan as it lies in memory -- the double at stamp_offset is PointStampId::stamp (16), the int32 at
 * scan->aux_offset the ring id (24), scan->index is ignored -- and dmsa_window_upload_from_ring with the static points as the tail of
 * globalPoints (xyz_static / ring_id_static / num_static of `p` are ignored). */
int dmsa_window_ring_push_aos(dmsa_ctx* ctx, const dmsa_aos_view* scan, int32_t stamp_offset);
int dmsa_window_upload_from_ring_aos(dmsa_ctx* ctx, const dmsa_window_problem* p, double t0, const dmsa_aos_view* static_points);
/* Everything a context allocates for problems up to these sizes, now: the first optimize call of a context otherwise pays tens of
 * milliseconds of hipMalloc.  max_table_rows = n_total (window) or the number of keyframes; max_params = 6 (poses - 1). */
int dmsa_reserve(dmsa_ctx* ctx, int64_t max_points, int32_t max_table_rows, int32_t max_params);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_AOS_H */
