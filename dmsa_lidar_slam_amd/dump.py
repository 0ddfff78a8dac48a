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
stage dump    'DMSAST03' | int32 model (1 window, 2 keyframes) | int32 P | int32 a | int32 M | int32 M_level1 | int32 table_rows | int64 Mm | int64 n |
              float32 table[table_rows x 12] ([R | t] row-major per pose-table row, the start poses as the optimiser sees them) |
              float32 global_xyz[4n] | float32 global_normal[4n] (keyframes only) | int32 seg_offset[M+1] | int32 members[Mm] |
              float32 info_mats[9M] (column-major) | float32 weights[M] |
              float32 fit_mean[3M] | float32 fit_cov[9M] (column-major, before limitCovariance) | float32 weights_raw[M] (pow(-1) of the counts)
              ('DMSAST01', the layout of round 4, lacks these three arrays) |
              float32 eig_values[3M] | float32 eig_vectors[9M] (eigensolver.eigenvalues().real() / eigenvectors().real() of fit_cov, column-major,
              Gaussians.h:184-188; 'DMSAST02', the layout of round 5, lacks these two) | float64 error_vec[M+a] | float64 jacobian[(M+a) x P] (column-major) |
              float64 H[P x P] (lambda on the diagonal) | float64 step_raw[P] | float64 step_clamped[P] | float64 error0 | int32 best_k | int32 pad |
              float64 params_after_line_search[P]
              -- iteration 0 of optimizeSet, stage by stage: written by oracle/ref_harness/ref_main.cpp from the REAL reference (a class derived
              from DmsaOptimizer reaches its protected members) and by the oracle (orc_stage_dump_*), compared in tests/test_ref_fixtures.py
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


def read_keyframe_map(path: str) -> MapManagement:
    """'DMSAKF01' back into a MapManagement (gravity rows as dumped, no odometry rows: the dump does not carry them)."""
    with open(path, "rb") as fh:
        assert fh.read(8) == b"DMSAKF01", path
        f, use_gravity, n, min_grid, _ = struct.unpack("<iiqff", fh.read(24))

        def arr(dt, *shape):
            count = int(np.prod(shape))
            return np.frombuffer(fh.read(count * np.dtype(dt).itemsize), dt).reshape(shape).copy()

        gravity = arr("<f8", 3)
        cov = arr("<f8", 3, 3).T.copy()
        bal = float(arr("<f8", 1)[0])
        ro, rt, off = arr("<f8", f, 3), arr("<f8", f, 3), arr("<i8", f + 1)
        xyz, nrm, ring = arr("<f4", n, 4), arr("<f4", n, 4), arr("<i4", n)
        mg, gp = arr("<f8", f, 3), arr("<i4", f)
    return MapManagement(relOrientations=ro, relTranslations=rt, frameOffsets=off, localPoints=xyz, localNormals=nrm, ringIds=ring, minGridSize=np.float32(min_grid),
                         useGravityErrorTerms=bool(use_gravity), measuredGravity=mg, gravityPlausible=gp, gravity=gravity, Cov_grav_inv=cov, balancingFactorGrav=bal)


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


def read_stage_dump(path: str) -> dict:
    """'DMSAST03' (or round 5's 'DMSAST02' / round 4's 'DMSAST01'): every intermediate result of iteration 0 of optimizeSet (see the module docstring)."""
    with open(path, "rb") as fh:
        magic = fh.read(8)
        assert magic in (b"DMSAST01", b"DMSAST02", b"DMSAST03"), path
        model, P, a, M, M1, rows = struct.unpack("<6i", fh.read(24))
        Mm, n = struct.unpack("<qq", fh.read(16))

        def arr(dt, *shape):
            count = int(np.prod(shape)) if shape else 1
            return np.frombuffer(fh.read(count * np.dtype(dt).itemsize), dt).reshape(shape).copy()

        d = dict(model=model, P=P, a=a, M=M, M1=M1, Mm=Mm, n=n)
        d["table"] = arr("<f4", rows, 12)
        d["global_xyz"] = arr("<f4", n, 4)
        d["global_normal"] = arr("<f4", n, 4) if model == 2 else None
        d["seg_offset"] = arr("<i4", M + 1)
        d["members"] = arr("<i4", Mm)
        d["info"] = arr("<f4", M, 9)
        d["weights"] = arr("<f4", M)
        v2, v3 = magic != b"DMSAST01", magic == b"DMSAST03"
        d["fit_mean"] = arr("<f4", M, 3) if v2 else None
        d["fit_cov"] = arr("<f4", M, 9) if v2 else None
        d["weights_raw"] = arr("<f4", M) if v2 else None
        d["eig_values"] = arr("<f4", M, 3) if v3 else None
        d["eig_vectors"] = arr("<f4", M, 9) if v3 else None
        d["error_vec"] = arr("<f8", M + a)
        d["jacobian"] = arr("<f8", P, M + a).T.copy()  # column-major (M+a) x P on disk
        d["H"] = arr("<f8", P, P).T.copy()
        d["step_raw"] = arr("<f8", P)
        d["step"] = arr("<f8", P)
        d["error0"] = float(arr("<f8", 1)[0])
        d["best_k"] = int(arr("<i4", 2)[0])
        d["params_after"] = arr("<f8", P)
        assert fh.read(1) == b"", "trailing bytes in " + path
    return d
