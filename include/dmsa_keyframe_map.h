/*
 * dmsa_keyframe_map.h — the seam that shards the keyframe pass: MapManagement::getSubmap / updatePosesFromSubmap
 * (include/DMSA/MapManagement.h:254-288) on plain arrays, plus the neighbourhood cut of the 8-GPU pass.
 *
 * Keyframe poses are RELATIVE poses (ConsecutivePoses.h:45-67).  getSubmap(from, to) cuts out an independent problem whose first
 * frame carries its global pose (and therefore stays fixed); updatePosesFromSubmap writes the optimised relative poses of columns
 * from+1 .. to back.  Host-only double arithmetic (no device, no context): "3 x n col-major" == Eigen::Matrix3Xd::data().
 * Return 0 or DMSA_ERR_INVALID (-1).
 */
#ifndef DMSA_KEYFRAME_MAP_H
#define DMSA_KEYFRAME_MAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* `parts` contiguous neighbourhoods [from_i, to_i] (inclusive) of a map of num_frames keyframes that share one boundary frame
 * (to_i == from_{i+1}), edges at round(i * (num_frames - 1) / parts); needs num_frames >= parts + 1.  Not a reference function: the
 * reference optimises one neighbourhood per new keyframe (DmsaSlam.h:212-238); this is the cut SURVEY.md 8(e) shards over GPUs. */
int dmsa_neighbourhood_ranges(int32_t num_frames, int32_t parts, int32_t* from_out, int32_t* to_out);

/* == MapManagement::getSubmap (MapManagement.h:254-276), pose part: n = to_id - from_id + 1 frames.
 * sub_rel_*: 3 x n col-major, column 0 = the GLOBAL pose of frame from_id.
 * odom_rel_transl (n x 3) / odom_rel_orient_mat (n x 9, each 3x3 col-major), optional: the odometry measurement every frame gets
 * when the submap is rebuilt through addKeyframe (:337-355) -- its relative pose at extraction time (row 0 = the global pose). */
int dmsa_submap_poses(int32_t num_frames, const double* rel_orient, const double* rel_transl, int32_t from_id, int32_t to_id, double* sub_rel_orient,
                      double* sub_rel_transl, double* odom_rel_transl, double* odom_rel_orient_mat);

/* == MapManagement::updatePosesFromSubmap (MapManagement.h:278-288): the submap's relative poses are re-derived from its global poses
 * (submap.keyframePoses.global2relative(), :280) and columns 1 .. n-1 overwrite columns from_id+1 .. to_id of the map (IN/OUT). */
int dmsa_update_poses_from_submap(int32_t num_frames, double* rel_orient, double* rel_transl, int32_t from_id, int32_t to_id,
                                  const double* sub_rel_orient, const double* sub_rel_transl);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_KEYFRAME_MAP_H */
