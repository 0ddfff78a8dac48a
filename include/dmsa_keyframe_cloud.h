/*
 * dmsa_keyframe_cloud.h — C ABI of keyframe creation ("keyframe creation (normals k = 6)"): the producers of
 * the keyframe pass's inputs (local PointNormal clouds + ring ids of KeyframeData).
 *
 *   DmsaSlam::updateNormals        include/DMSA/DmsaSlam.h:553-567   (pcl::NormalEstimationOMP, setKSearch(6), viewpoint = origin)
 *   DmsaSlam::addNewKeyframeToMap  include/DMSA/DmsaSlam.h:497-551   (thin globalPoints, move into the frame of control pose 0, normals)
 *   DmsaSlam::initializeMap        include/DMSA/DmsaSlam.h:469-495   (first keyframe: buffer cloud 0 as it is + normals)
 *
 * PCL pieces restated (recalled from PCL 1.10; its source is not in this container): FLANN exact k-NN with the query point as its own
 * first neighbour, flann::L2_Simple float distances, results in ascending distance (ties here: ascending index);
 * pcl::computeMeanAndCovarianceMatrix (single pass, float accumulators, neighbours in search order), pcl::eigen33 (scaled
 * characteristic polynomial, trigonometric roots, eigenvector = largest cross product of two rows of A - lambda I),
 * curvature = |lambda_0 / trace|, pcl::flipNormalTowardsViewpoint.  Fewer than 3 neighbours or a non-finite query give NaN.
 */
#ifndef DMSA_KEYFRAME_CLOUD_H
#define DMSA_KEYFRAME_CLOUD_H

#include "dmsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* == updateNormals(cloud, origin) (DmsaSlam.h:553-567).  normal_out n x 4 floats (normal_x, normal_y, normal_z, curvature);
 * nn_index_out (optional) n x k neighbour indices in search order, -1 padded.  k <= 8 (the reference uses 6).  cell_hint > 0 sizes
 * the search grid (cells of ~cell_hint; results do not depend on it, only speed does: use about twice the point spacing). */
int dmsa_update_normals(dmsa_ctx* ctx, const float* xyz, int64_t n, int32_t k, float cell_hint, const float viewpoint[3], float* normal_out,
                        int32_t* nn_index_out);

/* == the cloud part of addNewKeyframeToMap (DmsaSlam.h:497-531): randomGridDownsampling(globalPoints, minGridSize) with srand(seed),
 * p_local = currRotInv * (p - currWorldPose) in float for control pose 0 (pos0, orient0: axis-angle), ring ids of the kept points,
 * normals (k = 6, viewpoint = sensor origin).  Outputs sized by capacity (points); src_index_out = index into global_xyz of every
 * keyframe point.  DMSA_ERR_INVALID if capacity is too small (num_out still set).  global_xyz == NULL && ids == NULL: use the first n
 * global points and ring ids of the problem uploaded to this context (the window optimizeSet just finished) instead of host arrays. */
int dmsa_make_keyframe_cloud(dmsa_ctx* ctx, const float* global_xyz, const int32_t* ids, int64_t n, float min_grid_size, uint32_t seed, const double pos0[3],
                             const double orient0[3], float* xyz_local_out, float* normal_out, int32_t* ring_out, int32_t* src_index_out, int64_t capacity,
                             int64_t* num_out);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_KEYFRAME_CLOUD_H */
