"""Host-side mirror of the reference's problem interface for the DMSA hot path.

The reference seam is `DmsaOptimizer<PointT>::optimizeSet(OptimizablePointSet<PointT>&, DmsaOptimSettings)`
(DmsaOptimizer.h:54) with two concrete point sets:
  * ContinuousTrajectory (ContinuousTrajectory.h:24)  — sliding-window model, PointStampId
  * MapManagement        (MapManagement.h:20)         — keyframe-set model,  PointNormal
These classes carry exactly the state the hot path reads, as numpy arrays in the reference's
memory layouts (Eigen column-major 3xn doubles, PCL float[4] points), and hand it to the C ABI.
Names follow the reference (relativePoses, stamps, trajTime, tformIdPerPoint, ringIds, ...).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _capi as capi


@dataclass
class DmsaOptimSettings:
    """DmsaOptimizer.h:25-39 — same names, same defaults."""

    num_iter: int = 15
    epsilon: float = 1e-5
    use_analytic_jacobi: bool = False
    step_length_optim: float = 0.05
    max_step: float = 0.01
    gauss_split: bool = False
    grid_size_1_factor: float = 2.0
    grid_size_2_factor: float = 5.0
    min_num_points_per_set: int = 6
    min_num_gaussians: int = 30
    lambda_diag: float = 0.00001
    use_centralization: bool = True

    def to_c(self) -> capi.Settings:
        return capi.Settings(
            int(self.num_iter), float(self.epsilon), int(self.use_analytic_jacobi), float(self.step_length_optim),
            float(self.max_step), int(self.gauss_split), float(self.grid_size_1_factor), float(self.grid_size_2_factor),
            int(self.min_num_points_per_set), int(self.min_num_gaussians), float(self.lambda_diag),
            int(self.use_centralization),
        )

    @staticmethod
    def sliding_window(use_imu: bool = False, num_iter: int = 10) -> "DmsaOptimSettings":
        """Effective sliding-window settings of the reference pipeline (DmsaSlam.h:84-99, :455-466,
        config/slam_settings.yaml, Config.h:31-32; see SURVEY.md section 5 'wiring traps')."""
        s = DmsaOptimSettings(num_iter=num_iter, min_num_points_per_set=10)
        if use_imu:
            s.step_length_optim, s.max_step = 0.07, 0.05
        else:
            s.step_length_optim, s.max_step = 0.2, 0.3
        return s

    @staticmethod
    def keyframe_map(num_iter: int = 50) -> "DmsaOptimSettings":
        """Effective keyframe-pass settings (DmsaSlam.h:89-98, slam_settings.yaml)."""
        return DmsaOptimSettings(num_iter=num_iter, epsilon=1e-4, step_length_optim=0.2, max_step=0.01, gauss_split=True,
                                 grid_size_1_factor=2.0, min_num_points_per_set=10)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def _xyz4(a):
    a = np.asarray(a, dtype=np.float32)
    if a.ndim == 2 and a.shape[1] == 3:
        a = np.concatenate([a, np.ones((a.shape[0], 1), np.float32)], axis=1)
    assert a.ndim == 2 and a.shape[1] == 4
    return np.ascontiguousarray(a)


@dataclass
class ContinuousTrajectory:
    """Sliding-window problem.  Pose arrays are stored as (C, 3) C-contiguous == Eigen 3xC column-major."""

    relOrientations: np.ndarray   # (C,3) controlPoses.relativePoses.Orientations (axis-angle)
    relTranslations: np.ndarray   # (C,3)
    stamps: np.ndarray            # (C,)  controlPoses.stamps
    trajTime: np.ndarray          # (n_total,)
    localPoints: np.ndarray       # (N,4) float32
    tformIdPerPoint: np.ndarray   # (N,) int32
    ringIds: np.ndarray           # (N,) int32  (PointStampId::id)
    staticPoints: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.float32))
    staticRingIds: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.int32))
    minGridSize: float = 0.3
    useImuErrorTerms: bool = False
    dt_res: float = 0.001
    balancingImu: float = 0.001
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.805]))
    paramIndices: np.ndarray | None = None
    preintImuRots: np.ndarray | None = None      # (C,3,3) — stored transposed == column-major 3x3
    preintRelPositions: np.ndarray | None = None  # (C,3)
    preintRelVelocity: np.ndarray | None = None   # (C,3)
    CovPVRot_inv: np.ndarray | None = None        # (C,9,9) symmetric (layout-agnostic)

    def __post_init__(self):
        self.relOrientations = _f64(self.relOrientations)
        self.relTranslations = _f64(self.relTranslations)
        c = self.relOrientations.shape[0]
        assert self.relOrientations.shape == (c, 3) and self.relTranslations.shape == (c, 3)
        self.stamps = _f64(self.stamps, (c,))
        self.trajTime = _f64(self.trajTime)
        self.localPoints = _xyz4(self.localPoints)
        n = self.localPoints.shape[0]
        self.tformIdPerPoint = np.ascontiguousarray(self.tformIdPerPoint, dtype=np.int32)
        self.ringIds = np.ascontiguousarray(self.ringIds, dtype=np.int32)
        assert self.tformIdPerPoint.shape == (n,) and self.ringIds.shape == (n,)
        self.staticPoints = _xyz4(self.staticPoints) if len(self.staticPoints) else np.zeros((0, 4), np.float32)
        self.staticRingIds = np.ascontiguousarray(self.staticRingIds, dtype=np.int32)
        assert self.staticRingIds.shape == (self.staticPoints.shape[0],)
        self.gravity = _f64(self.gravity, (3,))
        if self.useImuErrorTerms:
            self.paramIndices = np.ascontiguousarray(self.paramIndices, dtype=np.int32)
            self.preintImuRots = _f64(self.preintImuRots, (c, 3, 3))
            self.preintRelPositions = _f64(self.preintRelPositions, (c, 3))
            self.preintRelVelocity = _f64(self.preintRelVelocity, (c, 3))
            self.CovPVRot_inv = _f64(self.CovPVRot_inv, (c, 9, 9))

    @property
    def numControlPoses(self) -> int:
        return self.relOrientations.shape[0]

    @property
    def numParams(self) -> int:
        return 6 * (self.numControlPoses - 1)

    def copy(self) -> "ContinuousTrajectory":
        import copy

        return copy.deepcopy(self)

    def getPoseParameters(self) -> np.ndarray:
        """Poses::getParamsAsVector (Poses.h:64-70): pose 0 excluded."""
        return np.concatenate([self.relOrientations[1:].ravel(), self.relTranslations[1:].ravel()])

    def to_c(self) -> capi.WindowProblem:
        p = capi.WindowProblem()
        p.num_control_poses = self.numControlPoses
        p.rel_orient = capi.ptr(self.relOrientations, C.c_double)
        p.rel_transl = capi.ptr(self.relTranslations, C.c_double)
        p.stamps = capi.ptr(self.stamps, C.c_double)
        p.n_total = self.trajTime.shape[0]
        p.traj_time = capi.ptr(self.trajTime, C.c_double)
        p.num_points = self.localPoints.shape[0]
        p.xyz_local = capi.ptr(self.localPoints, C.c_float)
        p.tform_idx = capi.ptr(self.tformIdPerPoint, C.c_int32)
        p.ring_id = capi.ptr(self.ringIds, C.c_int32)
        p.num_static = self.staticPoints.shape[0]
        p.xyz_static = capi.ptr(self.staticPoints, C.c_float)
        p.ring_id_static = capi.ptr(self.staticRingIds, C.c_int32)
        p.min_grid_size = float(self.minGridSize)
        p.use_imu = int(self.useImuErrorTerms)
        p.dt_res = float(self.dt_res)
        p.balancing_imu = float(self.balancingImu)
        p.gravity = (C.c_double * 3)(*self.gravity)
        if self.useImuErrorTerms:
            # (C,3,3) numpy row-major of R^T == column-major of R: callers store R itself, so transpose here
            self._rot_cm = np.ascontiguousarray(np.transpose(self.preintImuRots, (0, 2, 1)))
            self._cov_cm = np.ascontiguousarray(np.transpose(self.CovPVRot_inv, (0, 2, 1)))
            p.param_indices = capi.ptr(self.paramIndices, C.c_int32)
            p.preint_rot = capi.ptr(self._rot_cm, C.c_double)
            p.preint_pos = capi.ptr(self.preintRelPositions, C.c_double)
            p.preint_vel = capi.ptr(self.preintRelVelocity, C.c_double)
            p.cov_pvrot_inv = capi.ptr(self._cov_cm, C.c_double)
        return p


@dataclass
class MapManagement:
    """Keyframe-set problem (a submap as produced by MapManagement::getSubmap, MapManagement.h:254-276)."""

    relOrientations: np.ndarray   # (F,3) keyframePoses.relativePoses.Orientations
    relTranslations: np.ndarray   # (F,3)
    frameOffsets: np.ndarray      # (F+1,) int64 prefix of points per keyframe
    localPoints: np.ndarray       # (n,4) float32
    localNormals: np.ndarray      # (n,4) float32
    ringIds: np.ndarray           # (n,) int32
    minGridSize: float = 0.3
    useGravityErrorTerms: bool = False
    useOdometryErrorTerms: bool = False
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.805]))
    Cov_grav_inv: np.ndarray = field(default_factory=lambda: np.eye(3) / 0.3 ** 2)
    balancingFactorGrav: float = 1.0
    balancingFactorOdom: float = 1000.0
    measuredGravity: np.ndarray | None = None     # (F,3)
    gravityPlausible: np.ndarray | None = None    # (F,) int32
    odomRelTransl: np.ndarray | None = None       # (F,3)
    odomRelOrientMat: np.ndarray | None = None    # (F,3,3)
    odometryTranslCovInv: np.ndarray = field(default_factory=lambda: np.eye(3) / 0.01 ** 2)
    odometryOrientCovInv: np.ndarray = field(default_factory=lambda: np.eye(3) / 0.01 ** 2)
    gridSizes: np.ndarray | None = None           # (F,) KeyframeData::gridSize of every frame (minGridSize = their minimum, MapManagement.h:123-130)

    def __post_init__(self):
        self.relOrientations = _f64(self.relOrientations)
        self.relTranslations = _f64(self.relTranslations)
        f = self.relOrientations.shape[0]
        if self.gridSizes is not None:
            self.gridSizes = np.ascontiguousarray(self.gridSizes, dtype=np.float32)
            assert self.gridSizes.shape == (f,)
        assert self.relOrientations.shape == (f, 3) and self.relTranslations.shape == (f, 3)
        self.frameOffsets = np.ascontiguousarray(self.frameOffsets, dtype=np.int64)
        assert self.frameOffsets.shape == (f + 1,)
        self.localPoints = _xyz4(self.localPoints)
        n = self.localPoints.shape[0]
        assert n == int(self.frameOffsets[-1])
        ln = np.asarray(self.localNormals, dtype=np.float32)
        if ln.shape[1] == 3:
            ln = np.concatenate([ln, np.zeros((n, 1), np.float32)], axis=1)
        self.localNormals = np.ascontiguousarray(ln)
        self.ringIds = np.ascontiguousarray(self.ringIds, dtype=np.int32)
        self.gravity = _f64(self.gravity, (3,))
        self.Cov_grav_inv = _f64(self.Cov_grav_inv, (3, 3))
        if self.useGravityErrorTerms:
            self.measuredGravity = _f64(self.measuredGravity, (f, 3))
            self.gravityPlausible = np.ascontiguousarray(self.gravityPlausible, dtype=np.int32)
        if self.useOdometryErrorTerms:
            self.odomRelTransl = _f64(self.odomRelTransl, (f, 3))
            self.odomRelOrientMat = _f64(self.odomRelOrientMat, (f, 3, 3))

    @property
    def numFrames(self) -> int:
        return self.relOrientations.shape[0]

    @property
    def numParams(self) -> int:
        return 6 * (self.numFrames - 1)

    def copy(self) -> "MapManagement":
        import copy

        return copy.deepcopy(self)

    def getPoseParameters(self) -> np.ndarray:
        return np.concatenate([self.relOrientations[1:].ravel(), self.relTranslations[1:].ravel()])

    def to_c(self) -> capi.KeyframeProblem:
        p = capi.KeyframeProblem()
        p.num_frames = self.numFrames
        p.rel_orient = capi.ptr(self.relOrientations, C.c_double)
        p.rel_transl = capi.ptr(self.relTranslations, C.c_double)
        p.frame_offset = capi.ptr(self.frameOffsets, C.c_int64)
        p.xyz_local = capi.ptr(self.localPoints, C.c_float)
        p.normal_local = capi.ptr(self.localNormals, C.c_float)
        p.ring_id = capi.ptr(self.ringIds, C.c_int32)
        p.min_grid_size = float(self.minGridSize)
        p.use_gravity = int(self.useGravityErrorTerms)
        p.use_odometry = int(self.useOdometryErrorTerms)
        p.gravity = (C.c_double * 3)(*self.gravity)
        p.cov_grav_inv = (C.c_double * 9)(*self.Cov_grav_inv.T.ravel())
        p.balancing_grav = float(self.balancingFactorGrav)
        p.balancing_odom = float(self.balancingFactorOdom)
        if self.useGravityErrorTerms:
            p.measured_gravity = capi.ptr(self.measuredGravity, C.c_double)
            p.gravity_plausible = capi.ptr(self.gravityPlausible, C.c_int32)
        if self.useOdometryErrorTerms:
            self._odom_cm = np.ascontiguousarray(np.transpose(self.odomRelOrientMat, (0, 2, 1)))
            p.odom_rel_transl = capi.ptr(self.odomRelTransl, C.c_double)
            p.odom_rel_orient_mat = capi.ptr(self._odom_cm, C.c_double)
        p.odom_transl_cov_inv = (C.c_double * 9)(*_f64(self.odometryTranslCovInv, (3, 3)).T.ravel())
        p.odom_orient_cov_inv = (C.c_double * 9)(*_f64(self.odometryOrientCovInv, (3, 3)).T.ravel())
        return p

    def getSubmap(self, fromId: int, toId: int) -> "MapManagement":
        """MapManagement::getSubmap (MapManagement.h:254-276): independent problem over keyframes
        fromId..toId whose first frame carries its GLOBAL pose (and therefore stays fixed)."""
        # pose part through the C ABI (include/dmsa_keyframe_map.h: the same code a C++ host links)
        n = toId - fromId + 1
        ro, rt = np.zeros((n, 3)), np.zeros((n, 3))
        odom_t, odom_R_cm = np.zeros((n, 3)), np.zeros((n, 9))
        full_o, full_t = _f64(self.relOrientations), _f64(self.relTranslations)
        rc = capi.load_library().dmsa_submap_poses(self.numFrames, capi.ptr(full_o, C.c_double), capi.ptr(full_t, C.c_double), int(fromId), int(toId),
                                                   capi.ptr(ro, C.c_double), capi.ptr(rt, C.c_double), capi.ptr(odom_t, C.c_double), capi.ptr(odom_R_cm, C.c_double))
        if rc != 0:
            raise ValueError(f"getSubmap({fromId}, {toId}) on {self.numFrames} frames")
        a, b = int(self.frameOffsets[fromId]), int(self.frameOffsets[toId + 1])
        sl = slice(fromId, toId + 1)
        # The reference rebuilds the submap through addKeyframe (:266), which stores as each frame's odometry measurement the relative
        # pose derived from the CURRENT global poses (:337-355): at extraction the odometry residuals are exactly zero, i.e. a prior on
        # the current estimate, not on the odometry of the time the keyframe was created.
        odom_R = None
        if self.odomRelTransl is not None or self.useOdometryErrorTerms:
            odom_R = np.ascontiguousarray(np.transpose(odom_R_cm.reshape(n, 3, 3), (0, 2, 1)))  # col-major 3x3 -> [k][row][col]
        else:
            odom_t = None
        # minGridSize of the submap = smallest gridSize of ITS frames (:260-272)
        grid = float(self.minGridSize) if self.gridSizes is None else float(self.gridSizes[sl].min())
        return MapManagement(
            relOrientations=ro, relTranslations=rt, frameOffsets=self.frameOffsets[fromId:toId + 2] - a,
            localPoints=self.localPoints[a:b], localNormals=self.localNormals[a:b], ringIds=self.ringIds[a:b],
            minGridSize=grid, useGravityErrorTerms=self.useGravityErrorTerms,
            useOdometryErrorTerms=self.useOdometryErrorTerms, gravity=self.gravity, Cov_grav_inv=self.Cov_grav_inv,
            balancingFactorGrav=self.balancingFactorGrav, balancingFactorOdom=self.balancingFactorOdom,
            measuredGravity=None if self.measuredGravity is None else self.measuredGravity[sl],
            gravityPlausible=None if self.gravityPlausible is None else self.gravityPlausible[sl],
            odomRelTransl=odom_t, odomRelOrientMat=odom_R,
            odometryTranslCovInv=self.odometryTranslCovInv, odometryOrientCovInv=self.odometryOrientCovInv,
            gridSizes=None if self.gridSizes is None else self.gridSizes[sl],
        )

    def updatePosesFromSubmap(self, fromId: int, toId: int, submap: "MapManagement") -> None:
        """MapManagement::updatePosesFromSubmap (MapManagement.h:278-288): relative poses of
        columns fromId+1..toId are overwritten by the submap's."""
        sub_o, sub_t = _f64(submap.relOrientations), _f64(submap.relTranslations)
        rc = capi.load_library().dmsa_update_poses_from_submap(self.numFrames, capi.ptr(self.relOrientations, C.c_double), capi.ptr(self.relTranslations, C.c_double),
                                                               int(fromId), int(toId), capi.ptr(sub_o, C.c_double), capi.ptr(sub_t, C.c_double))
        if rc != 0:
            raise ValueError(f"updatePosesFromSubmap({fromId}, {toId}) on {self.numFrames} frames")

    @staticmethod
    def addKeyframe(keyframeMap: "MapManagement | None", position_w, orient_w, localPoints, localNormals, ringIds, gridSize: float,
                    measuredGravity=None, gravityPlausible: bool = True, maxNumKeyframes: int = 256, **settings) -> "MapManagement":
        """MapManagement::addKeyframe (MapManagement.h:311-390) as a pure function: the map with one more keyframe at the GLOBAL pose
        (position_w, orient_w).  Relative poses are re-derived from the global ones (global2relative, :337), the new frame's odometry
        measurement is its own relative pose (:339-355), a full ring buffer drops its oldest frame (:323-335), minGridSize follows the
        smallest keyframe gridSize.  `settings` (useOdometryErrorTerms, balancingFactorOdom, ...) apply when the map is created."""
        from .posemath import global2relative, relative2global

        pts, nrm = _xyz4(localPoints), np.asarray(localNormals, np.float32)
        pts[:, 3] = 1.0  # :357-358
        ids = np.ascontiguousarray(ringIds, np.int32)
        pos, ori = _f64(position_w, (3,)), _f64(orient_w, (3,))
        grav = np.zeros(3) if measuredGravity is None else _f64(measuredGravity, (3,))
        if keyframeMap is None:
            go, gt = ori[None, :], pos[None, :]
            offsets = np.array([0, pts.shape[0]], np.int64)
            cloud, normals, rings = pts, nrm, ids
            mg, gp = grav[None, :], np.array([int(gravityPlausible)], np.int32)
            odom_t, odom_R, grids = np.zeros((0, 3)), np.zeros((0, 3, 3)), np.array([gridSize], np.float32)
            base = dict(settings)
        else:
            m = keyframeMap
            go, gt = relative2global(m.relOrientations, m.relTranslations)  # :313
            first = 1 if m.numFrames >= maxNumKeyframes else 0               # full ring buffer: the oldest keyframe leaves
            a = int(m.frameOffsets[first])
            go, gt = np.vstack([go[first:], ori]), np.vstack([gt[first:], pos])
            offsets = np.concatenate([m.frameOffsets[first:] - a, [m.frameOffsets[-1] - a + pts.shape[0]]]).astype(np.int64)
            cloud, normals, rings = np.vstack([m.localPoints[a:], pts]), np.vstack([m.localNormals[a:], _pad4(nrm)]), np.concatenate([m.ringIds[a:], ids])
            z3, z33 = np.zeros((m.numFrames, 3)), np.tile(np.eye(3), (m.numFrames, 1, 1))
            mg = np.vstack([(m.measuredGravity if m.measuredGravity is not None else z3)[first:], grav])
            gp = np.concatenate([(m.gravityPlausible if m.gravityPlausible is not None else np.ones(m.numFrames, np.int32))[first:], [int(gravityPlausible)]]).astype(np.int32)
            odom_t, odom_R = (m.odomRelTransl if m.odomRelTransl is not None else z3)[first:], (m.odomRelOrientMat if m.odomRelOrientMat is not None else z33)[first:]
            old = m.gridSizes if m.gridSizes is not None else np.full(m.numFrames, m.minGridSize, np.float32)
            grids = np.concatenate([old[first:], [gridSize]]).astype(np.float32)  # an evicted frame no longer bounds minGridSize (:123-130)
            base = dict(useGravityErrorTerms=m.useGravityErrorTerms, useOdometryErrorTerms=m.useOdometryErrorTerms, gravity=m.gravity, Cov_grav_inv=m.Cov_grav_inv,
                        balancingFactorGrav=m.balancingFactorGrav, balancingFactorOdom=m.balancingFactorOdom, odometryTranslCovInv=m.odometryTranslCovInv,
                        odometryOrientCovInv=m.odometryOrientCovInv)
        ro, rt = global2relative(go, gt)  # :337
        from scipy.spatial.transform import Rotation as Rot

        odom_t = np.vstack([odom_t, rt[-1]])                                              # relativeTransl (:344 / :351)
        odom_R = np.concatenate([odom_R, Rot.from_rotvec(ro[-1]).as_matrix()[None]])      # relativeOrientMat = axang2rotm(relativeOrient) (:347)
        return MapManagement(relOrientations=ro, relTranslations=rt, frameOffsets=offsets, localPoints=cloud, localNormals=_pad4(normals), ringIds=rings,
                             minGridSize=float(grids.min()), measuredGravity=mg, gravityPlausible=gp, odomRelTransl=odom_t, odomRelOrientMat=odom_R,
                             gridSizes=grids, **base)


def _pad4(a):
    a = np.asarray(a, np.float32)
    return a if a.shape[1] == 4 else np.concatenate([a, np.zeros((a.shape[0], 1), np.float32)], axis=1)

