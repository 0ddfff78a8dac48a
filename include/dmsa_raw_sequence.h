/*
 * dmsa_raw_sequence.h — a flat dump of the messages the node reads from a rosbag, so that recorded sequences can be replayed without ROS.
 *
 * dmsa_slam_ros::spin (src/dmsa_slam_ros.cpp:240-307) walks its rosbag(s) in bag order and hands sensor_msgs/PointCloud2 messages of the
 * lidar topic to callbackPointCloud (:374-512) and sensor_msgs/Imu messages to callbackImuData (:309-320).  rosbag needs ROS, which this
 * library does not depend on; the dump keeps exactly what the two callbacks read, in the order the bag delivers it:
 *
 *   file     := "DMSARAW1" record*
 *   record   := u32 type (1 PointCloud2, 2 Imu) | u32 0 | u64 payload_bytes | payload
 *   PointCloud2 payload := f64 header stamp [s] | u32 height | u32 width | u32 point_step | u32 num_fields | u32 field_offset[num_fields]
 *                          | u64 data_bytes | data[data_bytes]                       (-> dmsa_pointcloud2 of dmsa_wire_formats.h)
 *   Imu payload         := f64 header stamp [s] | f64 angular_velocity[3] | f64 linear_acceleration[3]
 *
 * (little endian, no padding).  scripts/rosbag_to_raw.py writes such a file from a bag on a machine that has ROS; the reader and the
 * writer below are host code (no device needed).
 */
#ifndef DMSA_RAW_SEQUENCE_H
#define DMSA_RAW_SEQUENCE_H

#include "dmsa_wire_formats.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { DMSA_RAW_POINTCLOUD2 = 1, DMSA_RAW_IMU = 2 };
#define DMSA_RAW_END 1 /* dmsa_raw_next: no more records (a positive return value, not an error) */

typedef struct dmsa_raw_imu {
    double stamp;        /* msg->header.stamp.toSec()                                      */
    double ang_vel[3];   /* msg->angular_velocity   (callbackImuData :312-314)             */
    double lin_acc[3];   /* msg->linear_acceleration (:315-317; in g for acceleration_in_g sensors, the node scales) */
} dmsa_raw_imu;

typedef struct dmsa_raw_reader dmsa_raw_reader;
int dmsa_raw_open(const char* path, dmsa_raw_reader** out);
void dmsa_raw_close(dmsa_raw_reader* r);
/* Next record in file order.  *type_out = DMSA_RAW_POINTCLOUD2: *msg_out is filled (its pointers stay valid until the next call on this
 * reader; delta_t_pcs = stamp - previous point cloud's stamp, 0 for the first, as :394 computes it); DMSA_RAW_IMU: *imu_out is filled.
 * Returns DMSA_OK, DMSA_RAW_END, or DMSA_ERR_INVALID for a truncated / malformed file. */
int dmsa_raw_next(dmsa_raw_reader* r, int32_t* type_out, dmsa_pointcloud2* msg_out, dmsa_raw_imu* imu_out);

typedef struct dmsa_raw_writer dmsa_raw_writer;
int dmsa_raw_create(const char* path, dmsa_raw_writer** out);
int dmsa_raw_write_pointcloud2(dmsa_raw_writer* w, const dmsa_pointcloud2* msg);
int dmsa_raw_write_imu(dmsa_raw_writer* w, const dmsa_raw_imu* imu);
int dmsa_raw_finish(dmsa_raw_writer* w); /* flushes, closes and frees; DMSA_ERR_INVALID if a write failed */

#ifdef __cplusplus
}
#endif
#endif /* DMSA_RAW_SEQUENCE_H */
