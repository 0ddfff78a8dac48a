"""Python mirror of include/dmsa_raw_sequence.h: the flat dump of PointCloud2 / Imu messages that stands in for the rosbag loop of
dmsa_slam_ros::spin (src/dmsa_slam_ros.cpp:240-307).  Host code: needs the library, not a GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from .wire_formats import PointCloud2Msg

POINTCLOUD2, IMU, _END = 1, 2, 1


class RawSequenceError(RuntimeError):
    pass


class RawWriter:
    def __init__(self, path: str):
        self._lib = capi.load_library()
        self._w = C.c_void_p()
        if self._lib.dmsa_raw_create(str(path).encode(), C.byref(self._w)) != capi.DMSA_OK:
            raise RawSequenceError(f"cannot create {path}")

    def writePointCloud2(self, msg: PointCloud2Msg):
        cm = msg.to_c()
        if self._lib.dmsa_raw_write_pointcloud2(self._w, C.byref(cm)) != capi.DMSA_OK:
            raise RawSequenceError("write failed")

    def writeImu(self, stamp: float, ang_vel, lin_acc):
        m = capi.RawImu(float(stamp), (C.c_double * 3)(*[float(v) for v in ang_vel]), (C.c_double * 3)(*[float(v) for v in lin_acc]))
        if self._lib.dmsa_raw_write_imu(self._w, C.byref(m)) != capi.DMSA_OK:
            raise RawSequenceError("write failed")

    def close(self):
        if self._w:
            w, self._w = self._w, None
            if self._lib.dmsa_raw_finish(w) != capi.DMSA_OK:
                raise RawSequenceError("the dump could not be written completely")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class RawReader:
    """Iterates over the records in bag order: ("pointcloud2", PointCloud2Msg) or ("imu", (stamp, ang_vel, lin_acc))."""

    def __init__(self, path: str):
        self._lib = capi.load_library()
        self._r = C.c_void_p()
        if self._lib.dmsa_raw_open(str(path).encode(), C.byref(self._r)) != capi.DMSA_OK:
            raise RawSequenceError(f"{path} is not a DMSARAW1 dump")

    def __iter__(self):
        return self

    def __next__(self):
        kind, msg, imu = C.c_int32(0), capi.PointCloud2(), capi.RawImu()
        rc = self._lib.dmsa_raw_next(self._r, C.byref(kind), C.byref(msg), C.byref(imu))
        if rc == _END:
            raise StopIteration
        if rc != capi.DMSA_OK:
            raise RawSequenceError("truncated or malformed dump")
        if kind.value == IMU:
            return "imu", (imu.stamp, np.array(imu.ang_vel[:]), np.array(imu.lin_acc[:]))
        offs = np.ctypeslib.as_array(msg.field_offsets, shape=(msg.num_fields,)).copy() if msg.num_fields else np.zeros(0, np.uint32)
        data = np.ctypeslib.as_array(msg.data, shape=(msg.data_bytes,)).copy() if msg.data_bytes else np.zeros(0, np.uint8)
        return "pointcloud2", PointCloud2Msg(height=msg.height, width=msg.width, point_step=msg.point_step, field_offsets=offs, data=data, stamp=msg.stamp_msg)

    def close(self):
        if self._r:
            self._lib.dmsa_raw_close(self._r)
            self._r = None

    __del__ = close
