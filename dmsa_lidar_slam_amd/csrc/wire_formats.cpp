// wire_formats.cpp — host half of include/dmsa_wire_formats.h: TUM pose lines and the pose composition of OutputManagement.
// (The PointCloud2 decoder is a device kernel: k_decode_pointcloud2 in static_kernels.hip, glue in dmsa_api.cpp.)
#include "../../include/dmsa_wire_formats.h"

#include <cmath>
#include <cstdio>

#include "host_math.h"

using namespace dmsa;

namespace {
// Eigen's Quaterniond(Matrix3d) (QuaternionBase assignment from a rotation matrix): trace branch, else the largest diagonal entry
void quat_of(const Mat3& m, double q[4] /* x y z w */) {
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t, q[1] = (m(0, 2) - m(2, 0)) * t, q[2] = (m(1, 0) - m(0, 1)) * t;
        return;
    }
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m(k, j) - m(j, k)) * t;
    q[j] = (m(j, i) + m(i, j)) * t;
    q[k] = (m(k, i) + m(i, k)) * t;
}
}  // namespace

extern "C" {

int dmsa_format_tum_pose(double stamp, const double pos[3], const double orient[3], char* out, int32_t cap) {
    if (!pos || !orient || !out || cap < 1) return DMSA_ERR_INVALID;
    double q[4];
    quat_of(so3_exp({orient[0], orient[1], orient[2]}), q);
    // std::fixed + setprecision(p) prints like "%.pf"
    const int n = std::snprintf(out, (size_t)cap, "%.6f %.5f %.5f %.5f %.6f %.6f %.6f %.6f\n", stamp, pos[0], pos[1], pos[2], q[0], q[1], q[2], q[3]);
    if (n < 0 || n >= cap) return DMSA_ERR_INVALID;
    return n;
}

int dmsa_compose_nonkeyframe_pose(const double key_pos[3], const double key_orient[3], const double rel_transl[3], const double rel_orient[3], double pos_out[3],
                                  double orient_out[3]) {
    if (!key_pos || !key_orient || !rel_transl || !rel_orient || !pos_out || !orient_out) return DMSA_ERR_INVALID;
    const Mat3 keyRot = so3_exp({key_orient[0], key_orient[1], key_orient[2]});
    const Vec3 p = (keyRot * Vec3{rel_transl[0], rel_transl[1], rel_transl[2]}) + Vec3{key_pos[0], key_pos[1], key_pos[2]};
    const Vec3 o = so3_log(keyRot * so3_exp({rel_orient[0], rel_orient[1], rel_orient[2]}));
    pos_out[0] = p.x, pos_out[1] = p.y, pos_out[2] = p.z;
    orient_out[0] = o.x, orient_out[1] = o.y, orient_out[2] = o.z;
    return DMSA_OK;
}

}  // extern "C"
