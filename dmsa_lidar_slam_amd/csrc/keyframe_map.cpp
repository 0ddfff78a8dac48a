// keyframe_map.cpp — include/dmsa_keyframe_map.h: getSubmap / updatePosesFromSubmap on plain arrays (host double math).
#include "../../include/dmsa_keyframe_map.h"

#include <algorithm>
#include <cmath>

#include "../../include/dmsa_hip.h"
#include "host_math.h"

using namespace dmsa;

extern "C" {

int dmsa_neighbourhood_ranges(int32_t num_frames, int32_t parts, int32_t* from_out, int32_t* to_out) {
    if (parts < 1 || num_frames < parts + 1 || !from_out || !to_out) return DMSA_ERR_INVALID;
    const double step = (double)(num_frames - 1) / (double)parts;
    int32_t prev = 0;
    for (int32_t i = 0; i < parts; ++i) {
        const int32_t edge = i + 1 == parts ? num_frames - 1 : (int32_t)std::nearbyint((double)(i + 1) * step);  // round half to even
        from_out[i] = prev, to_out[i] = edge;
        prev = edge;
    }
    return DMSA_OK;
}

int dmsa_submap_poses(int32_t num_frames, const double* rel_orient, const double* rel_transl, int32_t from_id, int32_t to_id, double* sub_rel_orient,
                      double* sub_rel_transl, double* odom_rel_transl, double* odom_rel_orient_mat) {
    if (num_frames < 1 || !rel_orient || !rel_transl || from_id < 0 || to_id < from_id || to_id >= num_frames || !sub_rel_orient || !sub_rel_transl)
        return DMSA_ERR_INVALID;
    PoseChain full;
    full.resize(num_frames);
    std::copy(rel_orient, rel_orient + 3 * (size_t)num_frames, full.rel_o.begin());
    std::copy(rel_transl, rel_transl + 3 * (size_t)num_frames, full.rel_t.begin());
    full.relative_to_global();  // the global poses getSubmap reads (:262-263)
    const int n = to_id - from_id + 1;
    PoseChain sub;
    sub.resize(n);
    std::copy(full.glob_o.begin() + 3 * (size_t)from_id, full.glob_o.begin() + 3 * (size_t)(to_id + 1), sub.glob_o.begin());
    std::copy(full.glob_t.begin() + 3 * (size_t)from_id, full.glob_t.begin() + 3 * (size_t)(to_id + 1), sub.glob_t.begin());
    sub.global_to_relative();  // :273
    std::copy(sub.rel_o.begin(), sub.rel_o.end(), sub_rel_orient);
    std::copy(sub.rel_t.begin(), sub.rel_t.end(), sub_rel_transl);
    if (odom_rel_transl) std::copy(sub.rel_t.begin(), sub.rel_t.end(), odom_rel_transl);  // keyframeData.relativeTransl (:344)
    if (odom_rel_orient_mat)
        for (int k = 0; k < n; ++k) {  // relativeOrientMat = axang2rotm(relativeOrient) (:347)
            const Mat3 R = so3_exp({sub.rel_o[3 * k], sub.rel_o[3 * k + 1], sub.rel_o[3 * k + 2]});
            for (int c = 0; c < 3; ++c)
                for (int r = 0; r < 3; ++r) odom_rel_orient_mat[9 * (size_t)k + 3 * c + r] = R(r, c);
        }
    return DMSA_OK;
}

int dmsa_update_poses_from_submap(int32_t num_frames, double* rel_orient, double* rel_transl, int32_t from_id, int32_t to_id, const double* sub_rel_orient,
                                  const double* sub_rel_transl) {
    if (num_frames < 1 || !rel_orient || !rel_transl || from_id < 0 || to_id < from_id || to_id >= num_frames || !sub_rel_orient || !sub_rel_transl)
        return DMSA_ERR_INVALID;
    const int n = to_id - from_id + 1;
    PoseChain sub;
    sub.resize(n);
    std::copy(sub_rel_orient, sub_rel_orient + 3 * (size_t)n, sub.rel_o.begin());
    std::copy(sub_rel_transl, sub_rel_transl + 3 * (size_t)n, sub.rel_t.begin());
    sub.relative_to_global();   // the state optimizeSet leaves behind (setPoseParameters re-chains, MapManagement.h:197-202)
    sub.global_to_relative();   // :280
    for (int k = 1; k < n; ++k)
        for (int c = 0; c < 3; ++c) {
            rel_transl[3 * (size_t)(from_id + k) + c] = sub.rel_t[3 * (size_t)k + c];
            rel_orient[3 * (size_t)(from_id + k) + c] = sub.rel_o[3 * (size_t)k + c];
        }
    return DMSA_OK;
}

}  // extern "C"
