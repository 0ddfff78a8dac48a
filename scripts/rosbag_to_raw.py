#!/usr/bin/env python
"""rosbag -> flat dump (include/dmsa_raw_sequence.h) on a machine that has ROS 1 (`rosbag` Python package); NOT runnable in this
repository's build image (no ROS) and therefore untested here -- the format it writes is covered by tests/test_raw_sequence.py.

    python scripts/rosbag_to_raw.py <in.bag> <out.raw> --lidar-topic /hesai/pandar --imu-topic /alphasense/imu

Messages are written in bag order, like dmsa_slam_ros::spin reads them (src/dmsa_slam_ros.cpp:270-281): sensor_msgs/PointCloud2 of the
lidar topic (header stamp, height, width, point_step, field offsets, data) and sensor_msgs/Imu of the IMU topic (header stamp, angular
velocity, linear acceleration)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bag")
    ap.add_argument("out")
    ap.add_argument("--lidar-topic", required=True)
    ap.add_argument("--imu-topic", default=None)
    a = ap.parse_args()
    import rosbag  # ROS 1

    from dmsa_lidar_slam_amd.raw_sequence import RawWriter
    from dmsa_lidar_slam_amd.wire_formats import PointCloud2Msg

    topics = [a.lidar_topic] + ([a.imu_topic] if a.imu_topic else [])
    n_pc = n_imu = 0
    with RawWriter(a.out) as w, rosbag.Bag(a.bag) as bag:
        for topic, msg, _ in bag.read_messages(topics=topics):
            if topic == a.lidar_topic:
                offs = np.array([f.offset for f in msg.fields], np.uint32)
                w.writePointCloud2(PointCloud2Msg(height=msg.height, width=msg.width, point_step=msg.point_step, field_offsets=offs,
                                                  data=np.frombuffer(msg.data, np.uint8), stamp=msg.header.stamp.to_sec()))
                n_pc += 1
            else:
                av, la = msg.angular_velocity, msg.linear_acceleration
                w.writeImu(msg.header.stamp.to_sec(), [av.x, av.y, av.z], [la.x, la.y, la.z])
                n_imu += 1
    print(f"{a.out}: {n_pc} point clouds, {n_imu} IMU samples")


if __name__ == "__main__":
    main()
