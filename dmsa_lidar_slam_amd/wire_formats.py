"""Python mirror of the data formats either side of the path over include/dmsa_wire_formats.h: the per-sensor
sensor_msgs/PointCloud2 decoding of dmsa_slam_ros::callbackPointCloud (src/dmsa_slam_ros.cpp:374-486) on the device, and the TUM pose
lines of OutputManagement (OutputManagement.h:80-182) on the host."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi as capi
from .api import DmsaError


@dataclass
class PointCloud2Msg:
    """The parts of a sensor_msgs/PointCloud2 the callback reads."""
    height: int
    width: int
    point_step: int
    field_offsets: np.ndarray   # msg.fields[i].offset
    data: np.ndarray            # uint8 blob
    stamp: float                # msg.header.stamp.toSec()

    def to_c(self, delta_t_pcs: float = 0.0) -> capi.PointCloud2:
        self.field_offsets = np.ascontiguousarray(self.field_offsets, np.uint32)
        self.data = np.ascontiguousarray(self.data, np.uint8)
        m = capi.PointCloud2()
        m.height, m.width, m.point_step, m.num_fields = int(self.height), int(self.width), int(self.point_step), int(self.field_offsets.shape[0])
        m.field_offsets = capi.ptr(self.field_offsets, C.c_uint32)
        m.data, m.data_bytes = self.data.ctypes.data_as(C.POINTER(C.c_uint8)), int(self.data.shape[0])
        m.stamp_msg, m.delta_t_pcs = float(self.stamp), float(delta_t_pcs)
        return m


class PointCloud2Decoder:
    """callbackPointCloud (:374-486) for one config.sensor; keeps lastPcMsgStamp like the node (read by sensor "unknown" only)."""

    def __init__(self, sensor: str, device: int = 0):
        if sensor not in capi.SENSORS:
            raise ValueError(f"unknown sensor type {sensor!r}; one of {sorted(capi.SENSORS)}")
        self._lib = capi.load_library()
        self.sensor = sensor
        self.lastPcMsgStamp = -1.0
        ctx = C.c_void_p()
        rc = self._lib.dmsa_create(device, 0, C.byref(ctx))
        if rc != capi.DMSA_OK:
            raise DmsaError(f"dmsa_create failed with {rc}: the decoder runs on the GPU, there is no CPU fallback")
        self._ctx = ctx

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.dmsa_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    def decode(self, msg: PointCloud2Msg):
        """Returns (xyz (n,4) float32 with w = 0, stamps (n,) float64, ids (n,) int32), or None for the first message of an "unknown"
        sensor (the node only records its stamp, :388-392)."""
        if self.sensor == "unknown" and self.lastPcMsgStamp < 0.0:
            self.lastPcMsgStamp = msg.stamp
            return None
        delta = msg.stamp - self.lastPcMsgStamp  # :394 (lastPcMsgStamp is never updated afterwards in the reference)
        n = int(msg.height) * int(msg.width)
        xyz, st, ids = np.zeros((max(n, 1), 4), np.float32), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int32)
        cm = msg.to_c(delta)
        rc = self._lib.dmsa_decode_pointcloud2(self._ctx, C.byref(cm), capi.SENSORS[self.sensor], capi.ptr(xyz, C.c_float), capi.ptr(st, C.c_double),
                                               capi.ptr(ids, C.c_int32))
        if rc != capi.DMSA_OK:
            raise DmsaError(f"dmsa_decode_pointcloud2 failed with {rc}: {self._lib.dmsa_last_error(self._ctx).decode()}")
        return xyz[:n], st[:n], ids[:n]


def addPoseToFile(stamp: float, pos, orient) -> str:
    """OutputManagement::addPoseToFile (:80-96): one TUM line `stamp tx ty tz qx qy qz qw`."""
    lib = capi.load_library()
    p, o = np.ascontiguousarray(pos, np.float64), np.ascontiguousarray(orient, np.float64)
    buf = C.create_string_buffer(512)
    n = lib.dmsa_format_tum_pose(float(stamp), capi.ptr(p, C.c_double), capi.ptr(o, C.c_double), buf, 512)
    if n < 0:
        raise DmsaError(f"dmsa_format_tum_pose failed with {n}")
    return buf.raw[:n].decode()


def composeNonKeyframePose(keyframePos, keyframeOrient, Translation, Orientation):
    """saveDensePoses :148-153: pose of a non-keyframe scan from its pose relative to a keyframe."""
    lib = capi.load_library()
    a = [np.ascontiguousarray(v, np.float64) for v in (keyframePos, keyframeOrient, Translation, Orientation)]
    gp, go = np.zeros(3), np.zeros(3)
    rc = lib.dmsa_compose_nonkeyframe_pose(*[capi.ptr(v, C.c_double) for v in a], capi.ptr(gp, C.c_double), capi.ptr(go, C.c_double))
    if rc != capi.DMSA_OK:
        raise DmsaError(f"dmsa_compose_nonkeyframe_pose failed with {rc}")
    return gp, go
