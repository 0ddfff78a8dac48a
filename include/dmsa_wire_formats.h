/*
 * dmsa_wire_formats.h — C ABI of the data formats either side of the path: the per-sensor
 * sensor_msgs/PointCloud2 decoding of the node and the TUM pose lines it writes.
 *
 *   dmsa_slam_ros::callbackPointCloud   src/dmsa_slam_ros.cpp:374-486        (PointCloud2 bytes -> PointStampId)
 *   OutputManagement::addPoseToFile     include/DMSA/OutputManagement.h:80-96 (TUM line)
 *   OutputManagement::saveDensePoses    include/DMSA/OutputManagement.h:98-171 (non-keyframe pose composition :148-153, :176-182)
 *
 * rosbag / ROS message transport itself needs ROS (absent): the decoder takes the message's byte blob and the few header fields
 * it reads.  Decoding is byte work, one point per thread on the device; the pose text is host work.
 */
#ifndef DMSA_WIRE_FORMATS_H
#define DMSA_WIRE_FORMATS_H

#include "dmsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* config.sensor (dmsa_slam_ros.cpp:411-481) */
enum {
    DMSA_SENSOR_HESAI = 0,         /* stamp f64 @fields[4], ring u16 @fields[5]                                  (:411-419) */
    DMSA_SENSOR_OUSTER = 1,        /* t u32 ns @fields[4] relative to the header stamp, ring u8 @fields[6]       (:420-430) */
    DMSA_SENSOR_ROBOSENSE = 2,     /* ring u16 @fields[4], stamp f64 @fields[5]                                  (:431-439) */
    DMSA_SENSOR_VELODYNE = 3,      /* ring u16 @fields[4], time f32 @fields[5] relative to the header stamp      (:440-448) */
    DMSA_SENSOR_LIVOX_S = 4,       /* stamp f64 @fields[6] in seconds, id = k % 1000                             (:449-458) */
    DMSA_SENSOR_LIVOX_NS = 5,      /* stamp f64 @fields[6] in nanoseconds (1e-9 * value), id = k % 1000          (:459-469) */
    DMSA_SENSOR_SICK = 6,          /* time f32 @fields[8] relative to the header stamp, ring i8 @fields[11]      (:470-478) */
    DMSA_SENSOR_UNKNOWN = 7        /* stamp = header + deltaT * k / n, id = k % 1000                             (:479-486) */
};

/* The parts of a sensor_msgs/PointCloud2 the callback reads. */
typedef struct dmsa_pointcloud2 {
    uint32_t        height, width;   /* n = height * width points                                                        */
    uint32_t        point_step;      /* bytes per point                                                                  */
    uint32_t        num_fields;
    const uint32_t* field_offsets;   /* msg->fields[i].offset; fields 0..2 are x, y, z (float32)                         */
    const uint8_t*  data;            /* msg->data                                                                        */
    uint64_t        data_bytes;
    double          stamp_msg;       /* msg->header.stamp.toSec()                                                        */
    double          delta_t_pcs;     /* stampMsg - lastPcMsgStamp (:394), read by DMSA_SENSOR_UNKNOWN only               */
} dmsa_pointcloud2;

/* == the loop of callbackPointCloud (:399-486).  xyz_out n x 4 floats (w = 0: PointStampId is value-initialised, preProcess sets it
 * to 1 later), stamp_out n doubles (PointStampId::stamp), id_out n (PointStampId::id); isStatic = 0 for every point.
 * DMSA_ERR_INVALID when the sensor needs a field the message does not have or a field reaches beyond point_step / data. */
int dmsa_decode_pointcloud2(dmsa_ctx* ctx, const dmsa_pointcloud2* msg, int32_t sensor, float* xyz_out, double* stamp_out, int32_t* id_out);

/* == addPoseToFile (OutputManagement.h:80-96): "stamp tx ty tz qx qy qz qw\n" with 6 / 5 / 6 fixed decimals, the quaternion from
 * Eigen's Quaterniond(axang2rotm(orient)).  Writes at most cap bytes incl. the terminating 0; returns the line length (without the
 * 0) or a negative status. */
int dmsa_format_tum_pose(double stamp, const double pos[3], const double orient[3], char* out, int32_t cap);

/* == the non-keyframe pose composition of saveDensePoses / makeNonKeyframePoseGlobal (:148-153, :176-182):
 * pos_out = R(key_orient) * rel_transl + key_pos, orient_out = rotm2axang(R(key_orient) * R(rel_orient)). */
int dmsa_compose_nonkeyframe_pose(const double key_pos[3], const double key_orient[3], const double rel_transl[3], const double rel_orient[3],
                                  double pos_out[3], double orient_out[3]);

#ifdef __cplusplus
}
#endif
#endif /* DMSA_WIRE_FORMATS_H */
