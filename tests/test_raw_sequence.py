"""include/dmsa_raw_sequence.h: the flat dump of PointCloud2 / Imu messages standing in for the rosbag loop (src/dmsa_slam_ros.cpp:240-307).
Round trip of every field the two callbacks read, bag order, the deltaT_pcs the reader reconstructs (:394), and refusal of truncated files.
Host code: no GPU needed."""
import os

import numpy as np
import pytest

from dmsa_lidar_slam_amd import raw_sequence as rs
from dmsa_lidar_slam_amd.wire_formats import PointCloud2Msg


def _msg(rng, n, step, fields, stamp):
    return PointCloud2Msg(height=1, width=n, point_step=step, field_offsets=np.sort(rng.choice(step - 4, fields, replace=False)).astype(np.uint32),
                          data=rng.integers(0, 256, n * step, dtype=np.uint8), stamp=stamp)


def test_round_trip_in_bag_order(tmp_path):
    rng = np.random.default_rng(0)
    path = str(tmp_path / "seq.raw")
    sent = []
    with rs.RawWriter(path) as w:
        for k in range(7):
            if k % 3 == 1:
                m = _msg(rng, int(rng.integers(0, 300)), int(rng.integers(16, 48)), int(rng.integers(3, 9)), 1.6e9 + 0.1 * k)
                w.writePointCloud2(m)
                sent.append(("pointcloud2", m))
            else:
                imu = (1.6e9 + 0.1 * k + 0.01, rng.normal(size=3), rng.normal(size=3))
                w.writeImu(*imu)
                sent.append(("imu", imu))
    got = list(rs.RawReader(path))
    assert [k for k, _ in got] == [k for k, _ in sent]
    for (_, a), (kind, b) in zip(sent, got):
        if kind == "imu":
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        else:
            assert (a.height, a.width, a.point_step, a.stamp) == (b.height, b.width, b.point_step, b.stamp)
            assert np.array_equal(a.field_offsets, b.field_offsets) and np.array_equal(a.data, b.data)


def test_truncated_and_foreign_files_are_refused(tmp_path):
    rng = np.random.default_rng(1)
    path = str(tmp_path / "seq.raw")
    with rs.RawWriter(path) as w:
        w.writeImu(1.0, [1, 2, 3], [4, 5, 6])
        w.writePointCloud2(_msg(rng, 100, 32, 6, 2.0))
    blob = open(path, "rb").read()
    cut = str(tmp_path / "cut.raw")
    open(cut, "wb").write(blob[:-50])
    r = rs.RawReader(cut)
    assert next(r)[0] == "imu"
    with pytest.raises(rs.RawSequenceError):
        next(r)
    other = str(tmp_path / "other.bin")
    open(other, "wb").write(b"#ROSBAG V2.0\n" + bytes(64))
    with pytest.raises(rs.RawSequenceError):
        rs.RawReader(other)
    with pytest.raises(rs.RawSequenceError):
        rs.RawReader(str(tmp_path / "missing.raw"))
    empty = str(tmp_path / "empty.raw")
    rs.RawWriter(empty).close()
    assert list(rs.RawReader(empty)) == [] and os.path.getsize(empty) == 8
