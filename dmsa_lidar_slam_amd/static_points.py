"""Python mirror of the step before the DMSA hot path: DmsaSlam::addStaticPoints
(DmsaSlam.h:264-358) = candidate selection against the window cloud + isVisible (:360-375), randomGridDownsampling
(helpers.h:67-182, seeded) and getOverlap (:377-414), on the GPU through include/dmsa_static_points.h.  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi as capi
from .api import DmsaError


def _xyz4(a) -> np.ndarray:
    a = np.asarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] not in (3, 4):
        raise ValueError("points must be (n,3) or (n,4)")
    if a.shape[1] == 3:
        a = np.concatenate([a, np.ones((a.shape[0], 1), np.float32)], axis=1)
    return np.ascontiguousarray(a)


@dataclass
class StaticSelectProblem:
    """Inputs of the keyframe loop of addStaticPoints (names as in DmsaSlam.h:264-344)."""
    windowPoints: np.ndarray | None  # trajIn.globalPoints, (N,4) float32; None = the resident window of the shared optimizer context
    keyframeIds: np.ndarray         # closest keyframe ids that passed the distance gate, in closestKeyIds order
    frameOffsets: np.ndarray        # (K+1,) prefix of points per keyframe cloud
    keyPoints: np.ndarray           # (n,4) GLOBAL keyframe clouds, concatenated
    keyNormals: np.ndarray          # (n,4)
    keyRingIds: np.ndarray          # (n,)
    currPos: np.ndarray             # (3,) float32
    minGridSize: float
    numWindowResident: int = 0      # number of resident window points to use when windowPoints is None

    def __post_init__(self):
        self.windowPoints = _xyz4(self.windowPoints) if self.windowPoints is not None else None
        self.keyPoints, self.keyNormals = _xyz4(self.keyPoints), _xyz4(self.keyNormals)
        self.keyframeIds = np.ascontiguousarray(self.keyframeIds, dtype=np.int32)
        self.frameOffsets = np.ascontiguousarray(self.frameOffsets, dtype=np.int64)
        self.keyRingIds = np.ascontiguousarray(self.keyRingIds, dtype=np.int32)
        self.currPos = np.ascontiguousarray(self.currPos, dtype=np.float32)
        assert self.frameOffsets.shape == (self.keyframeIds.shape[0] + 1,) and self.frameOffsets[-1] == self.keyPoints.shape[0]
        assert self.keyNormals.shape == self.keyPoints.shape and self.keyRingIds.shape[0] == self.keyPoints.shape[0]

    def to_c(self) -> capi.StaticSelectProblem:
        p = capi.StaticSelectProblem()
        p.num_window = self.windowPoints.shape[0] if self.windowPoints is not None else int(self.numWindowResident)
        p.window_xyz = capi.ptr(self.windowPoints, C.c_float)
        p.num_keyframes = self.keyframeIds.shape[0]
        p.keyframe_ids = capi.ptr(self.keyframeIds, C.c_int32)
        p.frame_offset = capi.ptr(self.frameOffsets, C.c_int64)
        p.key_xyz, p.key_normal = capi.ptr(self.keyPoints, C.c_float), capi.ptr(self.keyNormals, C.c_float)
        p.key_ring = capi.ptr(self.keyRingIds, C.c_int32)
        p.cur_pos = (C.c_float * 3)(*[float(v) for v in self.currPos])
        p.min_grid_size = float(np.float32(self.minGridSize))
        return p


@dataclass
class StaticSelection:
    staticPoints: np.ndarray    # (m,4) in push_back order
    staticIds: np.ndarray       # (m,)
    overlapPerKeyframe: np.ndarray
    keyframeId: int
    minRelatedKeyId: int
    maxOverlap: int


class StaticPointSelector:
    """GPU implementation of the pieces of DmsaSlam::addStaticPoints (one context on one device)."""

    def __init__(self, device: int = 0, optimizer=None):
        """optimizer: a DmsaOptimizer whose context (and resident window cloud) this selector shares; None = own context."""
        self._lib = capi.load_library()
        self._owner = optimizer is None
        if optimizer is not None:
            self._ctx = optimizer._ctx
            return
        ctx = C.c_void_p()
        rc = self._lib.dmsa_create(device, 0, C.byref(ctx))
        if rc != capi.DMSA_OK:
            raise DmsaError(f"dmsa_create failed with {rc} (no usable HIP device: there is no CPU fallback)")
        self._ctx = ctx

    def close(self):
        if getattr(self, "_ctx", None) and self._owner:
            self._lib.dmsa_destroy(self._ctx)
        self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != capi.DMSA_OK:
            raise DmsaError(f"{what} failed with {rc}: {self._lib.dmsa_last_error(self._ctx).decode()}")

    def selectStaticPoints(self, prob: StaticSelectProblem) -> StaticSelection:
        cp = prob.to_c()
        cap = prob.keyPoints.shape[0]
        xyz, ids = np.zeros((max(cap, 1), 4), np.float32), np.zeros(max(cap, 1), np.int32)
        ov = np.zeros(max(1, prob.keyframeIds.shape[0]), np.int32)
        res = capi.StaticSelectResult()
        self._check(self._lib.dmsa_select_static_points(self._ctx, C.byref(cp), capi.ptr(xyz, C.c_float), capi.ptr(ids, C.c_int32), cap,
                                                        capi.ptr(ov, C.c_int32), C.byref(res)), "dmsa_select_static_points")
        m = res.num_static
        return StaticSelection(xyz[:m].copy(), ids[:m].copy(), ov[: prob.keyframeIds.shape[0]].copy(), res.keyframe_id, res.min_related_key_id, res.max_overlap)

    def getOverlap(self, pc1, pc2, maxDistOverlap: float, numResident: int = 0):
        """pc2 None: the first numResident resident window points of the shared optimizer context."""
        a = _xyz4(pc1)
        b = _xyz4(pc2) if pc2 is not None else None
        ov, nc = C.c_float(0.0), C.c_int64(0)
        self._check(self._lib.dmsa_get_overlap(self._ctx, capi.ptr(a, C.c_float), a.shape[0], capi.ptr(b, C.c_float), b.shape[0] if b is not None else int(numResident),
                                               float(np.float32(maxDistOverlap)), C.byref(ov), C.byref(nc)), "dmsa_get_overlap")
        return float(ov.value), int(nc.value)

    def randomGridDownsampling(self, rawPc, gridSize: float, seed: int) -> np.ndarray:
        """Indices into rawPc of the filtered cloud (one per occupied octree leaf, depth-first order), srand(seed)."""
        a = _xyz4(rawPc)
        out = np.zeros(max(1, a.shape[0]), np.int32)
        n = C.c_int64(0)
        self._check(self._lib.dmsa_random_grid_downsampling(self._ctx, capi.ptr(a, C.c_float), a.shape[0], float(np.float32(gridSize)), int(seed) & 0xFFFFFFFF,
                                                            capi.ptr(out, C.c_int32), a.shape[0], C.byref(n)), "dmsa_random_grid_downsampling")
        return out[: n.value].copy()

    def radiusExists(self, cloud, query, radius: float) -> np.ndarray:
        a, q = _xyz4(cloud), _xyz4(query)
        flags = np.zeros(max(1, q.shape[0]), np.uint8)
        self._check(self._lib.dmsa_radius_exists(self._ctx, capi.ptr(a, C.c_float), a.shape[0], capi.ptr(q, C.c_float), q.shape[0], float(np.float32(radius)),
                                                 flags.ctypes.data_as(C.POINTER(C.c_uint8))), "dmsa_radius_exists")
        return flags[: q.shape[0]].astype(bool)

    def preProcess(self, rawPc, seed: int, max_num_points_per_scan: int = 3000, minDistDS: float = 30.0, min_dist: float = 0.0, lidarToImuTform=None):
        """DmsaSlam::preProcess (DmsaSlam.h:569-634) on the coordinates of one scan; Config.h defaults.  Returns (filtered points
        n x 4 with w = 1 in the IMU frame, index into rawPc of each of them, gridSize of the filter pass that was kept)."""
        a = _xyz4(rawPc)
        cfg = capi.PreprocessConfig()
        cfg.max_num_points_per_scan, cfg.min_dist_ds, cfg.min_dist, cfg.seed = int(max_num_points_per_scan), float(minDistDS), float(min_dist), int(seed) & 0xFFFFFFFF
        T = np.eye(4, dtype=np.float32) if lidarToImuTform is None else np.asarray(lidarToImuTform, np.float32).reshape(4, 4)
        cfg.lidar_to_imu[:] = [float(v) for v in T.T.reshape(-1)]  # Eigen storage: column-major
        cap = a.shape[0]
        xyz, src = np.zeros((max(cap, 1), 4), np.float32), np.zeros(max(cap, 1), np.int32)
        n, grid = C.c_int64(0), C.c_float(0.0)
        self._check(self._lib.dmsa_preprocess_scan(self._ctx, capi.ptr(a, C.c_float), a.shape[0], C.byref(cfg), capi.ptr(xyz, C.c_float), capi.ptr(src, C.c_int32),
                                                   cap, C.byref(n), C.byref(grid)), "dmsa_preprocess_scan")
        return xyz[: n.value].copy(), src[: n.value].copy(), float(grid.value)

    def addStaticPoints(self, prob: StaticSelectProblem, seed: int):
        """The whole of DmsaSlam::addStaticPoints after the keyframe distance gate: selection, thinning at minGridSize/2
        (srand(seed)), overlap ratio against the window cloud.  Returns (selection, activePoints, activeIds, overlapToStatic)."""
        sel = self.selectStaticPoints(prob)
        if sel.staticPoints.shape[0] > 0:
            pick = self.randomGridDownsampling(sel.staticPoints, np.float32(prob.minGridSize) / np.float32(2.0), seed)
            active, active_ids = sel.staticPoints[pick], sel.staticIds[pick]
        else:
            active, active_ids = np.zeros((0, 4), np.float32), np.zeros(0, np.int32)
        overlap, _ = self.getOverlap(active, prob.windowPoints, prob.minGridSize, prob.numWindowResident)
        return sel, active, active_ids, overlap
