"""Python mirror of the producers of the hot path's inputs, over include/dmsa_window_setup.h:
ImuBuffer (ImuBuffer.h:14-175) and the setup half of ContinuousTrajectory (ContinuousTrajectory.h:228-568) as
DmsaSlam::prepareTrajectoryForOptimization drives it (DmsaSlam.h:416-461).  Names follow the reference.

Host arithmetic lives in the C++ library (window_setup.cpp); the per-point tformIdPerPoint search runs on the device.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _capi as capi
from .api import DmsaError
from .problems import ContinuousTrajectory


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _check(rc: int, what: str) -> None:
    if rc != capi.DMSA_OK:
        raise DmsaError(f"{what} failed with {rc}")


class ImuBuffer:
    """ImuBuffer.h:14-175."""

    def __init__(self, maxNumMeas: int = 10000):
        self._lib = capi.load_library()
        h = C.c_void_p()
        _check(self._lib.dmsa_imu_buffer_create(int(maxNumMeas), C.byref(h)), "dmsa_imu_buffer_create")
        self._h = h
        self.maxNumMeas = int(maxNumMeas)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dmsa_imu_buffer_destroy(self._h)
            self._h = None

    __del__ = close

    def addMeasurement(self, AccVec, AngVelVec, stamp: float) -> None:
        a, w = _f64(AccVec, (3,)), _f64(AngVelVec, (3,))
        _check(self._lib.dmsa_imu_buffer_add(self._h, capi.ptr(a, C.c_double), capi.ptr(w, C.c_double), float(stamp)), "dmsa_imu_buffer_add")

    def getClosestMeasurement(self, t: float):
        a, w, d = np.zeros(3), np.zeros(3), C.c_double(0.0)
        _check(self._lib.dmsa_imu_buffer_closest(self._h, float(t), capi.ptr(a, C.c_double), capi.ptr(w, C.c_double), C.byref(d)), "dmsa_imu_buffer_closest")
        return a, w, d.value

    def state(self):
        """(numUpdates, oldestIndex, bias_gyr, getLatestStamp(), getOldestStamp())"""
        n, o, b, lt, ot = C.c_int32(0), C.c_int32(0), np.zeros(3), C.c_double(0.0), C.c_double(0.0)
        _check(self._lib.dmsa_imu_buffer_state(self._h, C.byref(n), C.byref(o), capi.ptr(b, C.c_double), C.byref(lt), C.byref(ot)), "dmsa_imu_buffer_state")
        return n.value, o.value, b, lt.value, ot.value


@dataclass
class TrajectoryState:
    """What a ContinuousTrajectory carries from initTraj to optimizeSet (and, as oldTraj, into the next window).
    3 x n Eigen matrices are stored as (n, 3) C-contiguous arrays (same bytes)."""

    t0: float
    horizon: float
    dt_res: float
    n_total: int
    stamps: np.ndarray            # (C,) controlPoses.stamps
    trajTime: np.ndarray          # (n_total,)
    paramIndices: np.ndarray      # (C,) int32
    relOrientations: np.ndarray   # (C,3)
    relTranslations: np.ndarray
    globOrientations: np.ndarray
    globTranslations: np.ndarray
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.805]))  # :345
    useImuErrorTerms: bool = False
    accMeas: np.ndarray | None = None          # (n_total,3)
    angVelMeas: np.ndarray | None = None
    preintImuRots: np.ndarray | None = None    # (C,3,3) R itself (row-major numpy)
    preintRelPositions: np.ndarray | None = None
    preintRelVelocity: np.ndarray | None = None
    CovPVRot_inv: np.ndarray | None = None     # (C,9,9)
    preintPosComplHor: np.ndarray | None = None

    @property
    def numControlPoses(self) -> int:
        return int(self.stamps.shape[0])

    def to_c(self) -> capi.TrajState:
        s = capi.TrajState()
        s.t0, s.horizon, s.dt_res, s.n_total, s.num_control_poses = float(self.t0), float(self.horizon), float(self.dt_res), int(self.n_total), self.numControlPoses
        s.stamps, s.traj_time = capi.ptr(self.stamps, C.c_double), capi.ptr(self.trajTime, C.c_double)
        s.acc_meas, s.ang_vel_meas = capi.ptr(self.accMeas, C.c_double), capi.ptr(self.angVelMeas, C.c_double)
        s.gravity = (C.c_double * 3)(*self.gravity)
        s.rel_orient, s.rel_transl = capi.ptr(self.relOrientations, C.c_double), capi.ptr(self.relTranslations, C.c_double)
        s.glob_orient, s.glob_transl = capi.ptr(self.globOrientations, C.c_double), capi.ptr(self.globTranslations, C.c_double)
        return s


def new_state(t0, horizon, dt_res, n_total, stamps, trajTime, paramIndices, useImu) -> TrajectoryState:
    c = stamps.shape[0]
    z = lambda: np.zeros((c, 3))  # noqa: E731  (the reference leaves these unset; a fresh heap gives zeros)
    return TrajectoryState(float(t0), float(horizon), float(dt_res), int(n_total), stamps, trajTime, paramIndices, z(), z(), z(), z(), useImuErrorTerms=bool(useImu))


def assemble_problem(traj: TrajectoryState, clouds, tformIdPerPoint, balancingImu: float = 0.001) -> ContinuousTrajectory:
    """The bookkeeping half of registerPcBuffer (:228-238): concatenated clouds, minGridSize = min over the clouds' gridSize, and the
    setup fields of `traj` as the optimizeSet problem."""
    xyz = np.concatenate([np.asarray(c[0], np.float32)[:, :3] for c in clouds])
    ids = np.concatenate([np.asarray(c[2], np.int32) for c in clouds])
    min_grid = np.float32(np.finfo(np.float32).max)
    for c in clouds:
        min_grid = min(min_grid, np.float32(c[3]))
    kw = {}
    if traj.useImuErrorTerms:
        kw = dict(paramIndices=traj.paramIndices, preintImuRots=traj.preintImuRots, preintRelPositions=traj.preintRelPositions,
                  preintRelVelocity=traj.preintRelVelocity, CovPVRot_inv=traj.CovPVRot_inv, balancingImu=balancingImu)
    return ContinuousTrajectory(
        relOrientations=traj.relOrientations.copy(), relTranslations=traj.relTranslations.copy(), stamps=traj.stamps, trajTime=traj.trajTime,
        localPoints=np.concatenate([xyz, np.ones((xyz.shape[0], 1), np.float32)], axis=1), tformIdPerPoint=tformIdPerPoint, ringIds=ids,
        minGridSize=float(min_grid), useImuErrorTerms=traj.useImuErrorTerms, dt_res=traj.dt_res, gravity=traj.gravity, **kw)


class WindowSetup:
    """The calls of DmsaSlam::prepareTrajectoryForOptimization (DmsaSlam.h:416-461)."""

    def __init__(self, device: int = 0):
        self._lib = capi.load_library()
        self._device = device
        self._ctx = None

    def _context(self):
        if self._ctx is None:  # only registerPcBuffer needs the GPU
            ctx = C.c_void_p()
            rc = self._lib.dmsa_create(self._device, 0, C.byref(ctx))
            if rc != capi.DMSA_OK:
                raise DmsaError(f"dmsa_create failed with {rc}: the window setup has no CPU fallback for the per-point search")
            self._ctx = ctx
        return self._ctx

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.dmsa_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    # ContinuousTrajectory.h:301-346
    def initTraj(self, t_min: float, t_max: float, numControlPoses: int, useImu: bool, dtResIn: float) -> TrajectoryState:
        hor, n = C.c_double(0.0), C.c_int32(0)
        _check(self._lib.dmsa_traj_dims(float(t_min), float(t_max), float(dtResIn), C.byref(hor), C.byref(n)), "dmsa_traj_dims")
        tt, st, pi = np.zeros(n.value), np.zeros(int(numControlPoses)), np.zeros(int(numControlPoses), np.int32)
        _check(self._lib.dmsa_traj_grids(hor.value, float(dtResIn), n.value, int(numControlPoses), capi.ptr(tt, C.c_double), capi.ptr(st, C.c_double),
                                         capi.ptr(pi, C.c_int32)), "dmsa_traj_grids")
        return new_state(t_min, hor.value, dtResIn, n.value, st, tt, pi, useImu)

    # :348-364
    def transferImuMeasurements(self, traj: TrajectoryState, imuBuffer: ImuBuffer) -> float:
        traj.accMeas, traj.angVelMeas = np.zeros((traj.n_total, 3)), np.zeros((traj.n_total, 3))
        worst = C.c_double(0.0)
        _check(self._lib.dmsa_traj_transfer_imu(imuBuffer._h, traj.t0, capi.ptr(traj.trajTime, C.c_double), traj.n_total, capi.ptr(traj.accMeas, C.c_double),
                                                capi.ptr(traj.angVelMeas, C.c_double), C.byref(worst)), "dmsa_traj_transfer_imu")
        return worst.value

    # :518-568
    def updatePreintFactors(self, traj: TrajectoryState, gyr_cov, acc_cov) -> None:
        c = traj.numControlPoses
        g, a = _f64(np.asarray(gyr_cov).T, (3, 3)), _f64(np.asarray(acc_cov).T, (3, 3))  # column-major
        rot, pos, vel, cov, hor = np.zeros((c, 3, 3)), np.zeros((c, 3)), np.zeros((c, 3)), np.zeros((c, 9, 9)), np.zeros(3)
        _check(self._lib.dmsa_traj_preint_factors(traj.n_total, c, capi.ptr(traj.paramIndices, C.c_int32), traj.dt_res, capi.ptr(traj.accMeas, C.c_double),
                                                  capi.ptr(traj.angVelMeas, C.c_double), capi.ptr(g, C.c_double), capi.ptr(a, C.c_double), capi.ptr(rot, C.c_double),
                                                  capi.ptr(pos, C.c_double), capi.ptr(vel, C.c_double), capi.ptr(cov, C.c_double), capi.ptr(hor, C.c_double)),
               "dmsa_traj_preint_factors")
        traj.preintImuRots = np.ascontiguousarray(np.transpose(rot, (0, 2, 1)))  # column-major blocks -> R
        traj.CovPVRot_inv = np.ascontiguousarray(np.transpose(cov, (0, 2, 1)))
        traj.preintRelPositions, traj.preintRelVelocity, traj.preintPosComplHor = pos, vel, hor

    # :366-468
    def updateInitialGuess(self, isInitialized: bool, traj: TrajectoryState, oldTraj: TrajectoryState | None, useImu: bool) -> bool:
        flag = C.c_int32(int(bool(isInitialized)))
        cur = traj.to_c()
        old = oldTraj.to_c() if oldTraj is not None else None
        _check(self._lib.dmsa_traj_update_initial_guess(C.byref(flag), C.byref(cur), C.byref(old) if old is not None else None, int(bool(useImu))),
               "dmsa_traj_update_initial_guess")
        return bool(flag.value)

    # :593-601
    def getSubmapGravityEstimate(self, traj: TrajectoryState) -> np.ndarray:
        """measuredGravity of a keyframe made from this window (needs updatePreintFactors and current GLOBAL control poses)."""
        out = np.zeros(3)
        cs = traj.to_c()
        _check(self._lib.dmsa_traj_submap_gravity_estimate(C.byref(cs), capi.ptr(traj.preintPosComplHor, C.c_double), capi.ptr(out, C.c_double)),
               "dmsa_traj_submap_gravity_estimate")
        return out

    # :228-261 (the per-point search; runs on the device)
    def tformIdPerPoint(self, traj: TrajectoryState, pointStamps) -> np.ndarray:
        st = _f64(pointStamps)
        out = np.zeros(max(1, st.shape[0]), np.int32)
        rc = self._lib.dmsa_traj_tform_indices(self._context(), capi.ptr(st, C.c_double), st.shape[0], traj.t0, capi.ptr(traj.trajTime, C.c_double), traj.n_total,
                                               capi.ptr(out, C.c_int32))
        if rc != capi.DMSA_OK:
            raise DmsaError(f"dmsa_traj_tform_indices failed with {rc}: {self._lib.dmsa_last_error(self._ctx).decode()}")
        return out[: st.shape[0]]

    def registerPcBuffer(self, traj: TrajectoryState, clouds, balancingImu: float = 0.001) -> ContinuousTrajectory:
        """clouds: the ring buffer in chronological order, each (xyz (n,3|4) float32, stamps (n,) float64, ids (n,) int32, gridSize).
        Returns the problem optimizeSet takes."""
        stamps = np.concatenate([_f64(c[1]) for c in clouds])
        return assemble_problem(traj, clouds, self.tformIdPerPoint(traj, stamps), balancingImu)

    def prepareTrajectoryForOptimization(self, clouds, oldTraj: TrajectoryState | None, isInitialized: bool, numControlPoses: int, dt_res: float,
                                         imuBuffer: ImuBuffer | None = None, cov_gyr=None, cov_acc=None, balancingImu: float = 0.001):
        """DmsaSlam.h:416-461.  Returns (traj state, problem for optimizeSet, isInitialized)."""
        t_min = min(float(np.min(c[1])) for c in clouds)  # pcBuffer->getMinMaxPointStamps
        t_max = max(float(np.max(c[1])) for c in clouds)
        use_imu = imuBuffer is not None
        traj = self.initTraj(t_min, t_max, numControlPoses, use_imu, dt_res)
        if use_imu:
            self.transferImuMeasurements(traj, imuBuffer)
            self.updatePreintFactors(traj, cov_gyr, cov_acc)
        isInitialized = self.updateInitialGuess(isInitialized, traj, oldTraj, use_imu)
        return traj, self.registerPcBuffer(traj, clouds, balancingImu), isInitialized
