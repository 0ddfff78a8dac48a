"""Synthetic sensor_msgs/PointCloud2 blobs with the field layouts of the drivers the node supports (dmsa_slam_ros.cpp:411-481),
built with numpy structured dtypes — an encoder that is independent of both decoders under test."""
import numpy as np

from dmsa_lidar_slam_amd.wire_formats import PointCloud2Msg

# (name, numpy type) in message order; offsets follow from packing + explicit padding like the real drivers
LAYOUTS = {
    # hesai: x y z intensity timestamp(f64) ring(u16)
    "hesai": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("timestamp", "<f8"), ("ring", "<u2")],
    # ouster: x y z _pad intensity t(u32) reflectivity(u16) ring(u8) ambient(u16) range(u32)
    "ouster": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad0", "<f4"), ("intensity", "<f4"), ("t", "<u4"), ("reflectivity", "<u2"), ("ring", "u1"),
               ("pad1", "u1"), ("ambient", "<u2"), ("pad2", "<u2"), ("range", "<u4")],
    # robosense: x y z intensity ring(u16) timestamp(f64)
    "robosense": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("pad0", "<u2"), ("timestamp", "<f8")],
    # velodyne: x y z intensity ring(u16) time(f32)
    "velodyne": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("ring", "<u2"), ("time", "<f4")],  # time unaligned at byte 18
    # livox XYZRTLT: x y z reflectivity(f32) tag(u8) line(u8) timestamp(f64)
    "livoxXYZRTLT_s": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "<f4"), ("tag", "u1"), ("line", "u1"), ("timestamp", "<f8")],
    "livoxXYZRTLT_ns": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("reflectivity", "<f4"), ("tag", "u1"), ("line", "u1"), ("timestamp", "<f8")],
    # sick multiscan: 12 fields, time f32 at field 8, layer i8 at field 11
    "sick": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("i", "<f4"), ("range", "<f4"), ("azimuth", "<f4"), ("elevation", "<f4"), ("t_hi", "<u4"), ("ts", "<f4"),
             ("echo", "i1"), ("reflector", "i1"), ("layer", "i1")],
    "unknown": [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4")],
}
# which message field (by index into msg.fields, pads excluded) carries stamp / ring
_PADS = ("pad0", "pad1", "pad2")


def make_msg(sensor: str, n: int, seed: int = 0, stamp: float = 1.6e9 + 12.25, height: int = 1, truncate_step: int = 0):
    """Returns (PointCloud2Msg, expected dict with x y z stamp id)."""
    rng = np.random.default_rng(seed)
    fields = LAYOUTS[sensor]
    dt = np.dtype(fields)  # packed (align=False): offsets exactly as listed
    rec = np.zeros(n, dt)
    for name, _ in fields:
        kind = rec.dtype[name].kind
        rec[name] = rng.normal(0, 20, n) if kind == "f" else rng.integers(0, 100, n)
    k = np.arange(n, dtype=np.uint32)
    exp = {"x": rec["x"].copy(), "y": rec["y"].copy(), "z": rec["z"].copy()}
    if sensor in ("hesai", "robosense"):
        rec["timestamp"] = stamp + np.sort(rng.uniform(0, 0.1, n))
        rec["ring"] = rng.integers(0, 128, n)
        exp["stamp"], exp["id"] = rec["timestamp"].copy(), rec["ring"].astype(np.int32)
    elif sensor == "ouster":
        rec["t"] = np.sort(rng.integers(0, 100_000_000, n)).astype(np.uint32)
        rec["ring"] = rng.integers(0, 128, n)
        exp["stamp"], exp["id"] = stamp + 1e-9 * rec["t"].astype(np.float64), rec["ring"].astype(np.int32)
    elif sensor == "velodyne":
        rec["time"] = np.sort(rng.uniform(-0.1, 0.0, n)).astype(np.float32)
        rec["ring"] = rng.integers(0, 64, n)
        exp["stamp"], exp["id"] = stamp + rec["time"].astype(np.float64), rec["ring"].astype(np.int32)
    elif sensor == "livoxXYZRTLT_s":
        rec["timestamp"] = stamp + np.sort(rng.uniform(0, 0.1, n))
        exp["stamp"], exp["id"] = rec["timestamp"].copy(), (k % 1000).astype(np.int32)
    elif sensor == "livoxXYZRTLT_ns":
        rec["timestamp"] = np.floor((stamp + np.sort(rng.uniform(0, 0.1, n))) * 1e9)
        exp["stamp"], exp["id"] = 1e-9 * rec["timestamp"], (k % 1000).astype(np.int32)
    elif sensor == "sick":
        rec["ts"] = np.sort(rng.uniform(0, 0.05, n)).astype(np.float32)
        rec["layer"] = rng.integers(-8, 8, n)
        exp["stamp"], exp["id"] = stamp + rec["ts"].astype(np.float64), rec["layer"].astype(np.int32)
    offsets = np.array([dt.fields[name][1] for name, _ in fields if name not in _PADS], np.uint32)
    data = np.frombuffer(rec.tobytes(), np.uint8).copy()
    w = n // height
    msg = PointCloud2Msg(height=height, width=w, point_step=dt.itemsize - truncate_step, field_offsets=offsets, data=data, stamp=stamp)
    return msg, exp


def expected_unknown(n, stamp, delta):
    k = np.arange(n, dtype=np.float64)
    return stamp + delta * k / float(n), (np.arange(n) % 1000).astype(np.int32)
