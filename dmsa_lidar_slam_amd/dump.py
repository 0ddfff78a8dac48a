"""Flat binary dumps of the two problem models (the stand-in for rosbag input): everything a C / C++ host needs to
call the C ABI without Python.  Little-endian, arrays in the ABI's own layouts (include/dmsa_hip.h).

keyframe map  'DMSAKF01' | int32 F | int32 use_gravity | int64 n | float32 min_grid_size | float32 pad |
              float64 gravity[3] | float64 cov_grav_inv[9] (col-major) | float64 balancing_grav |
              float64 rel_orient[3F] | float64 rel_transl[3F] | int64 frame_offset[F+1] |
              float32 xyz_local[4n] | float32 normal_local[4n] | int32 ring_id[n] |
              float64 measured_gravity[3F] | int32 gravity_plausible[F]
window        'DMSAWN01' | int32 C | int32 n_total | int64 N | int64 S | float32 min_grid_size | float32 pad | int32 use_imu | int32 pad |
              float64 dt_res | float64 rel_orient[3C] | float64 rel_transl[3C] | float64 stamps[C] | float64 traj_time[n_total] |
              float32 xyz_local[4N] | int32 tform_idx[N] | int32 ring_id[N] | float32 xyz_static[4S] | int32 ring_id_static[S]
              (IMU rows are not dumped: use_imu must be 0 for the reference harness)
poses         'DMSAPO01' | int32 F | int32 pad | float64 rel_orient[3F] | float64 rel_transl[3F]
"""
from __future__ import annotations

import struct

import numpy as np

from .problems import ContinuousTrajectory, MapManagement


def write_keyframe_map(path: str, m: MapManagement) -> None:
    f, n = m.numFrames, m.localPoints.shape[0]
    mg = m.measuredGravity if m.measuredGravity is not None else np.zeros((f, 3))
    gp = m.gravityPlausible if m.gravityPlausible is not None else np.ones(f, np.int32)
    with open(path, "wb") as fh:
        fh.write(b"DMSAKF01")
        fh.write(struct.pack("<iiqff", f, int(m.useGravityErrorTerms), n, float(m.minGridSize), 0.0))
        fh.write(np.asarray(m.gravity, "<f8").tobytes())
        fh.write(np.asarray(m.Cov_grav_inv, "<f8").T.copy().tobytes())  # col-major
        fh.write(struct.pack("<d", float(m.balancingFactorGrav)))
        for a, dt in ((m.relOrientations, "<f8"), (m.relTranslations, "<f8"), (m.frameOffsets, "<i8"), (m.localPoints, "<f4"), (m.localNormals, "<f4"),
                      (m.ringIds, "<i4"), (mg, "<f8"), (gp, "<i4")):
            fh.write(np.ascontiguousarray(a, dt).tobytes())


def write_window_problem(path: str, w: ContinuousTrajectory) -> None:
    c, nt, n, ns = w.numControlPoses, w.trajTime.shape[0], w.localPoints.shape[0], w.staticPoints.shape[0]
    with open(path, "wb") as fh:
        fh.write(b"DMSAWN01")
        fh.write(struct.pack("<iiqqffiid", c, nt, n, ns, float(w.minGridSize), 0.0, int(w.useImuErrorTerms), 0, float(w.dt_res)))
        for a, dt in ((w.relOrientations, "<f8"), (w.relTranslations, "<f8"), (w.stamps, "<f8"), (w.trajTime, "<f8"), (w.localPoints, "<f4"),
                      (w.tformIdPerPoint, "<i4"), (w.ringIds, "<i4"), (w.staticPoints, "<f4"), (w.staticRingIds, "<i4")):
            fh.write(np.ascontiguousarray(a, dt).tobytes())


def read_poses(path: str):
    with open(path, "rb") as fh:
        assert fh.read(8) == b"DMSAPO01"
        f, _ = struct.unpack("<ii", fh.read(8))
        ro = np.frombuffer(fh.read(24 * f), "<f8").reshape(f, 3).copy()
        rt = np.frombuffer(fh.read(24 * f), "<f8").reshape(f, 3).copy()
    return ro, rt
