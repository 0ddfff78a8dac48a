/*
 * dmsa_static_points.h — C ABI of the step immediately BEFORE the DMSA hot path in every window:
 * static-point selection, random grid thinning and overlap ratio of DmsaSlam::addStaticPoints.
 *
 *   DmsaSlam::addStaticPoints      include/DMSA/DmsaSlam.h:264-358
 *   DmsaSlam::isVisible            include/DMSA/DmsaSlam.h:360-375
 *   DmsaSlam::getOverlap           include/DMSA/DmsaSlam.h:377-414
 *   randomGridDownsampling         include/DMSA/helpers.h:67-182
 *   DmsaSlam::preProcess           include/DMSA/DmsaSlam.h:569-634   (the per-scan filter in front of the window, row f2)
 *
 * Same conventions as dmsa_hip.h (contexts, status codes, float[n][4] points, no CPU fallback).
 *
 * RESIDENT WINDOW CLOUD: wherever a function takes the window cloud (window_xyz of the selection, pc1 / pc2 of getOverlap, the
 * input of randomGridDownsampling), a NULL pointer means "the first n global points of the problem uploaded to this context, as
 * the last dmsa_transform_points / dmsa_optimize_* left them" -- the 1.3 M-point cloud then never crosses PCIe between the hot
 * path and the steps around it.  DMSA_ERR_INVALID when nothing is uploaded or n exceeds the resident point count.  The reference answers its
 * "nearest neighbour within minGridSize" questions with a FLANN kd-tree (pcl::KdTreeFLANN, flann::L2_Simple); only the
 * comparison `squared distance of the nearest neighbour <= radius^2` is ever used, which is the order-independent predicate
 * "some point lies within the radius" -- evaluated here on a uniform cell grid in HBM with the same float distance
 * ((dx*dx + dy*dy) + dz*dz, the accumulation order of L2_Simple).
 */
#ifndef DMSA_STATIC_POINTS_H
#define DMSA_STATIC_POINTS_H

#include "dmsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Inputs of the selection loop of addStaticPoints (DmsaSlam.h:300-344). */
typedef struct dmsa_static_select_problem {
    int64_t        num_window;      /* trajIn.globalPoints.size()                                                      */
    const float*   window_xyz;      /* N x 4, the window cloud the kd-tree is built on (:288)                          */
    int32_t        num_keyframes;   /* closest keyframes that passed the distance gate (:298-304), in closestKeyIds order */
    const int32_t* keyframe_ids;    /* K, the ring-buffer ids k (reported back as keyframeId / minRelatedKeyId)        */
    const int64_t* frame_offset;    /* K+1 prefix of points per keyframe cloud                                         */
    const float*   key_xyz;         /* n x 4, GLOBAL keyframe clouds (getGlobalKeyframeCloud, MapManagement.h:290-299) */
    const float*   key_normal;      /* n x 4, global normals                                                           */
    const int32_t* key_ring;        /* n, keyframeDataBuffer.at(k).ringIds(j)                                          */
    float          cur_pos[3];      /* currPosf = controlPoses.globalPoses.Translations.col(0).cast<float>() (:267-268) */
    float          min_grid_size;   /* trajIn.minGridSize; gate = (1.0f * minGridSize)^2 (:295)                        */
} dmsa_static_select_problem;

typedef struct dmsa_static_select_result {
    int64_t num_static;             /* staticPoints->size() before thinning                                            */
    int32_t keyframe_id;            /* keyframeId: keyframe with the largest overlap, first one wins, 0 if none (:270, :338-342) */
    int32_t min_related_key_id;     /* minRelatedKeyId: smallest id that contributed a point, -1 if none (:274, :333-334) */
    int32_t max_overlap;            /* maxOverlapKey                                                                   */
    int32_t pad;
} dmsa_static_select_result;

/* == the keyframe loop of addStaticPoints (DmsaSlam.h:300-344).  static_xyz_out (capacity x 4 floats, w = 1) / static_id_out
 * (capacity) receive the selected points in the reference's push_back order (keyframes in the given order, points ascending);
 * overlap_per_keyframe (K, optional) receives currOverlap of every keyframe.  Returns DMSA_ERR_INVALID if capacity is too
 * small (num_static is still reported). */
int dmsa_select_static_points(dmsa_ctx* ctx, const dmsa_static_select_problem* p, float* static_xyz_out, int32_t* static_id_out,
                              int64_t capacity, int32_t* overlap_per_keyframe, dmsa_static_select_result* res);

/* == getOverlap(pc1, pc2, maxDistOverlap) (DmsaSlam.h:377-414): fraction of the pc2 points that have a pc1 point within
 * maxDistOverlap; 0 when either cloud is empty.  num_corresp_out is optional. */
int dmsa_get_overlap(dmsa_ctx* ctx, const float* pc1_xyz, int64_t n1, const float* pc2_xyz, int64_t n2, float max_dist_overlap,
                     float* overlap_out, int64_t* num_corresp_out);

/* == randomGridDownsampling(rawPc, filteredPc, gridSize) (helpers.h:67-182) with srand(seed) in place of srand(time(0)):
 * one point per occupied PCL-octree leaf, leaves in depth-first order, the point chosen by glibc rand() exactly like
 * `id = int((double)rand() / RAND_MAX * (double)(indices.size() - 1))`.  picked_index_out (capacity) receives the index INTO
 * rawPc of every filtered point; num_out = octree.getLeafCount(). */
int dmsa_random_grid_downsampling(dmsa_ctx* ctx, const float* xyz, int64_t n, float grid_size, uint32_t seed,
                                  int32_t* picked_index_out, int64_t capacity, int64_t* num_out);

/* Predicate behind both kd-tree uses: flag_out[i] = 1 iff some cloud point lies within `radius` of query i
 * (squared L2_Simple distance <= radius*radius in float).  Non-finite cloud points never match, non-finite queries get 0. */
int dmsa_radius_exists(dmsa_ctx* ctx, const float* cloud_xyz, int64_t n_cloud, const float* query_xyz, int64_t n_query, float radius,
                       uint8_t* flag_out);

/* Knobs of DmsaSlam::preProcess that live in Config (Config.h:24-25, :38, :58). */
typedef struct dmsa_preprocess_config {
    int32_t  max_num_points_per_scan; /* Config.h:24 (3000; 1000 for livox, dmsa_slam_ros.cpp:204)                      */
    float    min_dist_ds;             /* Config.h:25 minDistDS: lower bound of the range threshold                      */
    float    min_dist;                /* Config.h:38: points at or below this range are dropped                         */
    uint32_t seed;                    /* srand(time(0)) of every filter pass -> srand(seed)                             */
    float    lidar_to_imu[16];        /* Config.h:58 lidarToImuTform in Eigen's storage order (column-major)            */
} dmsa_preprocess_config;

/* == DmsaSlam::preProcess(rawPc, filteredPc) (DmsaSlam.h:569-634) on the coordinates of one scan:
 *   1. adaptive random grid filter: randomGridDownsampling at 0.4 m, then 0.3 / 0.2 / 0.15 m while the result has fewer than
 *      max_num_points_per_scan points (:572-592); *grid_size_out = filteredPc->gridSize of the pass that was kept;
 *   2. ranges = Vector3f(x, y, z).norm(), thresRange = max(sorted ranges[min(max_num_points_per_scan, size - 1)], minDistDS) (:595-609);
 *   3. keep the points with min_dist < range < thresRange in their order (:611-623);
 *   4. pcl::transformPointCloud with lidarToImuTform, then data[3] = 1 (:626-630).
 * xyz_out (capacity x 4 floats) receives the filtered points, src_index_out (capacity) the index INTO the raw scan each of them came
 * from (stamp / id / isStatic of PointStampId travel with the point, PointStampId.h:33-45).  A scan without finite points gives
 * num_out = 0 (the reference would read rangesSorted[-1]).  Returns DMSA_ERR_INVALID if capacity is too small (num_out still set). */
int dmsa_preprocess_scan(dmsa_ctx* ctx, const float* raw_xyz, int64_t n, const dmsa_preprocess_config* cfg, float* xyz_out, int32_t* src_index_out,
                         int64_t capacity, int64_t* num_out, float* grid_size_out);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_STATIC_POINTS_H */
