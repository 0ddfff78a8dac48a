"""Python face of the C ABI (include/dmsa_hip.h) — mirrors the reference's optimizer interface.

    DmsaOptimizer().optimizeSet(pointSet, settings)      == DmsaOptimizer<PointT>::optimizeSet (DmsaOptimizer.h:54)

`pointSet` is a problems.ContinuousTrajectory (sliding window) or problems.MapManagement (keyframe set); its
relative poses are updated in place, exactly like the reference mutates the OptimizablePointSet it is given.
All compute happens in libdmsa_hip.so on the GPU; there is no CPU fallback (a missing library or device raises).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from .problems import ContinuousTrajectory, DmsaOptimSettings, MapManagement


class DmsaError(RuntimeError):
    pass


class DmsaOptimizer:
    """One context == one GPU == one host thread (the reference's optimizer is not re-entrant either)."""

    def __init__(self, device: int = 0, pose_table_host: bool = False, fixed_iters: bool = False, stage_timers: bool = False, debug: dict | None = None):
        """The reference's summation order (bit-identical to the CPU restatement) with pose tables built on the device -- the library has one
        path.  `debug`: switches of include/dmsa_debug.h by name, e.g. {"device_loop": 0} -- alternative implementations with the same
        results (tests, A/B timing)."""
        self._lib = capi.load_library()
        self._ctx = C.c_void_p()
        flags = (capi.FLAG_POSE_TABLE_HOST if pose_table_host else 0) | (capi.FLAG_FIXED_ITERS if fixed_iters else 0)
        flags |= capi.FLAG_STAGE_TIMERS if stage_timers else 0
        if debug:
            opts = capi.DebugOptions()
            self._lib.dmsa_default_debug_options(C.byref(opts))
            for k, v in debug.items():
                if not hasattr(opts, k):
                    raise KeyError(f"unknown debug switch {k!r} (include/dmsa_debug.h)")
                setattr(opts, k, int(v))
            rc = self._lib.dmsa_create_ex(int(device), flags, C.byref(opts), C.byref(self._ctx))
        else:
            rc = self._lib.dmsa_create(int(device), flags, C.byref(self._ctx))
        if rc != capi.DMSA_OK:
            self._ctx = None
            raise DmsaError(f"dmsa_create(device={device}) failed with {rc} (no usable HIP device; there is no CPU fallback)")
        self._problem = None
        self._cprob = None

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.dmsa_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != capi.DMSA_OK:
            msg = self._lib.dmsa_last_error(self._ctx)
            raise DmsaError(f"{what} failed with {rc}: {msg.decode() if msg else ''}")

    # -- the drop-in call ---------------------------------------------------------------------------
    def optimizeSet(self, pointSetToOptimize, settings: DmsaOptimSettings | None = None) -> capi.Report:
        settings = settings or DmsaOptimSettings()
        cs = settings.to_c()
        rep = capi.Report()
        cp = pointSetToOptimize.to_c()
        if isinstance(pointSetToOptimize, ContinuousTrajectory):
            rc = self._lib.dmsa_optimize_window(self._ctx, C.byref(cp), C.byref(cs), C.byref(rep))
        elif isinstance(pointSetToOptimize, MapManagement):
            rc = self._lib.dmsa_optimize_keyframes(self._ctx, C.byref(cp), C.byref(cs), C.byref(rep))
        else:
            raise TypeError("pointSetToOptimize must be a ContinuousTrajectory or a MapManagement")
        self._check(rc, "optimizeSet")
        self._problem, self._cprob = pointSetToOptimize, cp
        return rep

    # -- include/dmsa_aos.h: the reference's own point containers -------------------------------------------------------
    POINT_STAMP_ID = np.dtype({"names": ["x", "y", "z", "w", "stamp", "id", "isStatic"], "formats": ["<f4", "<f4", "<f4", "<f4", "<f8", "<i4", "<i4"],
                               "offsets": [0, 4, 8, 12, 16, 24, 28], "itemsize": 32})   # PointStampId.h:33-45
    POINT_NORMAL = np.dtype({"names": ["x", "y", "z", "w", "normal_x", "normal_y", "normal_z", "nw", "curvature"],
                             "formats": ["<f4"] * 9, "offsets": [0, 4, 8, 12, 16, 20, 24, 28, 32], "itemsize": 48})  # pcl::PointNormal

    @staticmethod
    def _view(arr, aux_offset, index):
        v = capi.AosView()
        v.base, v.count, v.stride, v.xyz_offset, v.aux_offset = arr.ctypes.data, arr.shape[0], arr.dtype.itemsize, 0, aux_offset
        v.index = index.ctypes.data_as(C.POINTER(C.c_int32)) if index is not None else None
        return v

    def optimizeSetAos(self, pointSetToOptimize, settings: DmsaOptimSettings | None = None, reserve: bool = False):
        """optimizeSet with the points handed over as the reference holds them: one pcl::PointCloud<PointStampId> per scan of the ring buffer
        (32-byte points) plus the static tail of globalPoints, or one pcl::PointCloud<PointNormal> per keyframe (48-byte points).  The
        structured arrays built here stand in for cloud.points.data(); returns (report, clouds) -- after the call the first array
        holds... nothing new: the global points are fetched with globalPointsAos()."""
        settings = settings or DmsaOptimSettings()
        cs, rep, p = settings.to_c(), capi.Report(), pointSetToOptimize
        cp = p.to_c()
        keep = []
        if isinstance(p, ContinuousTrajectory):
            off = getattr(p, "scanOffsets", None)
            if off is None:
                off = np.array([0, p.localPoints.shape[0]])
            views = (capi.AosView * (len(off) - 1))()
            for k in range(len(off) - 1):
                a, b = int(off[k]), int(off[k + 1])
                cloud = np.zeros(b - a, self.POINT_STAMP_ID)
                cloud["x"], cloud["y"], cloud["z"], cloud["w"] = p.localPoints[a:b, 0], p.localPoints[a:b, 1], p.localPoints[a:b, 2], 1.0
                cloud["id"], cloud["stamp"] = p.ringIds[a:b], 1.6e9
                idx = np.ascontiguousarray(p.tformIdPerPoint[a:b], np.int32)
                keep += [cloud, idx]
                views[k] = self._view(cloud, 24, idx)
            stat = np.zeros(p.staticPoints.shape[0], self.POINT_STAMP_ID)
            stat["x"], stat["y"], stat["z"], stat["w"] = p.staticPoints[:, 0], p.staticPoints[:, 1], p.staticPoints[:, 2], 1.0
            stat["id"], stat["stamp"], stat["isStatic"] = p.staticRingIds, -1000.0, 1
            sv = self._view(stat, 24, None)
            keep.append(stat)
            if reserve:
                self._check(self._lib.dmsa_reserve(self._ctx, p.localPoints.shape[0] + stat.shape[0], p.trajTime.shape[0], p.numParams), "reserve")
            # what integration/DmsaOptimizerHip.h hands over: NO flat point arrays -- the AoS calls must not look at them
            cp.num_points, cp.xyz_local, cp.tform_idx, cp.ring_id = 0, None, None, None
            cp.num_static, cp.xyz_static, cp.ring_id_static = 0, None, None
            rc = self._lib.dmsa_optimize_window_aos(self._ctx, C.byref(cp), views, len(off) - 1, C.byref(sv), C.byref(cs), C.byref(rep))
        else:
            off = p.frameOffsets
            views = (capi.AosView * p.numFrames)()
            for k in range(p.numFrames):
                a, b = int(off[k]), int(off[k + 1])
                cloud = np.zeros(b - a, self.POINT_NORMAL)
                cloud["x"], cloud["y"], cloud["z"], cloud["w"] = p.localPoints[a:b, 0], p.localPoints[a:b, 1], p.localPoints[a:b, 2], p.localPoints[a:b, 3]
                cloud["normal_x"], cloud["normal_y"], cloud["normal_z"], cloud["nw"] = (p.localNormals[a:b, c] for c in range(4))
                idx = np.ascontiguousarray(p.ringIds[a:b], np.int32)
                keep += [cloud, idx]
                views[k] = self._view(cloud, 16, idx)
            cp.xyz_local, cp.normal_local, cp.ring_id, cp.frame_offset = None, None, None, None
            rc = self._lib.dmsa_optimize_keyframes_aos(self._ctx, C.byref(cp), views, p.numFrames, C.byref(cs), C.byref(rep))
        self._check(rc, "optimizeSetAos")
        self._problem, self._cprob = p, cp
        return rep

    def ringPushAos(self, xyz_local, stamps, ring_ids):
        """dmsa_window_ring_push_aos: the scan as a pcl::PointCloud<PointStampId> (32-byte points) instead of three flat arrays."""
        cloud = np.zeros(len(stamps), self.POINT_STAMP_ID)
        cloud["x"], cloud["y"], cloud["z"], cloud["w"] = xyz_local[:, 0], xyz_local[:, 1], xyz_local[:, 2], 1.0
        cloud["stamp"], cloud["id"] = stamps, ring_ids
        v = self._view(cloud, 24, None)
        self._check(self._lib.dmsa_window_ring_push_aos(self._ctx, C.byref(v), 16), "window_ring_push_aos")

    def uploadFromRingAos(self, window: ContinuousTrajectory, t0: float):
        """dmsa_window_upload_from_ring_aos: the static points as the PointStampId tail of globalPoints."""
        cp = window.to_c()
        stat = np.zeros(window.staticPoints.shape[0], self.POINT_STAMP_ID)
        stat["x"], stat["y"], stat["z"], stat["w"] = window.staticPoints[:, 0], window.staticPoints[:, 1], window.staticPoints[:, 2], 1.0
        stat["id"], stat["stamp"], stat["isStatic"] = window.staticRingIds, -1000.0, 1
        sv = self._view(stat, 24, None)
        cp.num_points, cp.xyz_local, cp.tform_idx, cp.ring_id = 0, None, None, None
        cp.num_static, cp.xyz_static, cp.ring_id_static = 0, None, None
        self._check(self._lib.dmsa_window_upload_from_ring_aos(self._ctx, C.byref(cp), float(t0), C.byref(sv)), "window_upload_from_ring_aos")
        self._problem, self._cprob = window, cp

    def globalPointsAos(self, keyframes: bool = False) -> np.ndarray:
        """The final updateGlobalPoints written into a strided globalPoints container (PointStampId / PointNormal layout)."""
        n = self._num_points()
        out = np.zeros(n, self.POINT_NORMAL if keyframes else self.POINT_STAMP_ID)
        if not keyframes:
            out["stamp"], out["id"] = 7.0, 123  # fields the call must leave alone
        self._check(self._lib.dmsa_get_global_points_aos(self._ctx, out.ctypes.data, n, out.dtype.itemsize, 0, 16 if keyframes else -1), "get_global_points_aos")
        return out

    def powMinusOne(self, counts) -> np.ndarray:
        """Test hook: pow(-1) of member counts as the fit computes it on the device (the host libm's powf through its difference table)."""
        n = np.ascontiguousarray(counts, np.int32)
        out = np.zeros(n.shape[0], np.float32)
        self._check(self._lib.dmsa_debug_pow_minus_one(self._ctx, capi.ptr(n, C.c_int32), n.shape[0], capi.ptr(out, C.c_float)), "debug_pow_minus_one")
        return out

    def limitCovariance(self, cov):
        """Test hook: Gaussians::limitCovariance on an (n, 3, 3) array [row, col] as the fit computes it on the device.  Returns the limited
        covariances, eigenvalues().real(), eigenvectors().real() [row, col], Francis QR steps and info of the EigenSolver behind it."""
        A = np.ascontiguousarray(np.asarray(cov, np.float32).reshape(-1, 3, 3))
        n = A.shape[0]
        At = np.ascontiguousarray(A.transpose(0, 2, 1))
        out, Vt = np.zeros_like(At), np.zeros_like(At)
        ev = np.zeros((n, 3), np.float32)
        it, info = np.zeros(n, np.int32), np.zeros(n, np.int32)
        self._check(self._lib.dmsa_debug_limit_covariance(self._ctx, capi.ptr(At, C.c_float), n, capi.ptr(out, C.c_float), capi.ptr(ev, C.c_float),
                                                          capi.ptr(Vt, C.c_float), capi.ptr(it, C.c_int32), capi.ptr(info, C.c_int32)), "debug_limit_covariance")
        return np.ascontiguousarray(out.transpose(0, 2, 1)), ev, np.ascontiguousarray(Vt.transpose(0, 2, 1)), it, info

    def optimizeResident(self, settings: DmsaOptimSettings) -> capi.Report:
        """optimizeSet on the problem already resident in HBM (after upload() or a previous optimizeSet)."""
        cs = settings.to_c()
        rep = capi.Report()
        self._check(self._lib.dmsa_optimize_resident(self._ctx, C.byref(cs), C.byref(rep)), "optimize_resident")
        return rep

    def adaptiveStepSize(self, params, step, error0: float):
        """DmsaOptimizer::adaptiveStepSize (DmsaOptimizer.h:152-182) on the resident problem and its current Gaussians -> (parameters chosen, best_k)."""
        p = np.ascontiguousarray(params, np.float64).copy()
        st = np.ascontiguousarray(step, np.float64)
        k = C.c_int32()
        self._check(self._lib.dmsa_adaptive_step_size(self._ctx, capi.ptr(p, C.c_double), capi.ptr(st, C.c_double), float(error0), C.byref(k)), "adaptive_step_size")
        return p, int(k.value)

    def debugCounters(self) -> dict:
        """include/dmsa_debug.h: dmsa_debug_counters (retries after a timed-out device-side wait / a wrong sort-width guess, pairs the
        Jacobian batches left out because they equal evaluation 0)."""
        c = capi.DebugCounters()
        self._check(self._lib.dmsa_get_debug_counters(self._ctx, C.byref(c)), "get_debug_counters")
        return {n: int(getattr(c, n)) for n, _ in c._fields_}

    def lastError(self) -> str:
        msg = self._lib.dmsa_last_error(self._ctx)
        return msg.decode() if msg else ""

    def poses(self):
        """Current relative poses (n,3),(n,3) of the resident problem."""
        p = self._problem
        n = p.relOrientations.shape[0]
        ro, rt = np.zeros((n, 3)), np.zeros((n, 3))
        self._check(self._lib.dmsa_get_poses(self._ctx, capi.ptr(ro, C.c_double), capi.ptr(rt, C.c_double)), "get_poses")
        return ro, rt

    def globalPoints(self) -> np.ndarray:
        n = self._num_points()
        out = np.zeros((n, 4), np.float32)
        self._check(self._lib.dmsa_get_global_points(self._ctx, capi.ptr(out, C.c_float), n), "get_global_points")
        return out

    # -- stage-level calls (parity tests, benchmark) ---------------------------------------------------
    def upload(self, pointSet):
        cp = pointSet.to_c()
        if isinstance(pointSet, ContinuousTrajectory):
            rc = self._lib.dmsa_window_upload(self._ctx, C.byref(cp))
        else:
            rc = self._lib.dmsa_keyframes_upload(self._ctx, C.byref(cp))
        self._check(rc, "upload")
        self._problem, self._cprob = pointSet, cp

    # -- include/dmsa_window_ring.h: the window's scans resident in HBM -------------------------------------------------
    def ringCreate(self, num_scans: int, max_points_per_scan: int, max_static_points: int, max_n_total: int, max_control_poses: int = 6):
        cfg = capi.WindowRingConfig(int(num_scans), int(max_points_per_scan), int(max_static_points), int(max_n_total), int(max_control_poses))
        self._check(self._lib.dmsa_window_ring_create(self._ctx, C.byref(cfg)), "window_ring_create")

    def ringPush(self, xyz_local, stamps, ring_ids):
        """pcBuffer->addElem(scan) (RingBuffer.h:67-88): the newest scan replaces the oldest one in HBM."""
        x = np.ascontiguousarray(xyz_local, np.float32)
        if x.shape[1] == 3:
            x = np.ascontiguousarray(np.concatenate([x, np.ones((x.shape[0], 1), np.float32)], axis=1))
        t = np.ascontiguousarray(stamps, np.float64)
        r = np.ascontiguousarray(ring_ids, np.int32)
        assert x.shape[0] == t.size == r.size
        self._check(self._lib.dmsa_window_ring_push(self._ctx, capi.ptr(x, C.c_float), capi.ptr(t, C.c_double), capi.ptr(r, C.c_int32), x.shape[0]), "window_ring_push")

    def ringPoints(self):
        s, n = C.c_int32(0), C.c_int64(0)
        self._check(self._lib.dmsa_window_ring_points(self._ctx, C.byref(s), C.byref(n)), "window_ring_points")
        return int(s.value), int(n.value)

    def uploadFromRing(self, window: ContinuousTrajectory, t0: float):
        """The resident scans + this window's control poses / time grid / static points become the problem (registerPcBuffer on the device).
        `window.localPoints / tformIdPerPoint / ringIds` are not read."""
        cp = window.to_c()
        cp.num_points = 0
        self._check(self._lib.dmsa_window_upload_from_ring(self._ctx, C.byref(cp), float(t0)), "window_upload_from_ring")
        self._problem, self._cprob = window, cp

    def _num_points(self) -> int:
        p = self._problem
        if isinstance(p, ContinuousTrajectory):
            return p.localPoints.shape[0] + p.staticPoints.shape[0]
        return p.localPoints.shape[0]

    def centralize(self):
        self._check(self._lib.dmsa_centralize(self._ctx), "centralize")

    def decentralize(self):
        self._check(self._lib.dmsa_decentralize(self._ctx), "decentralize")

    def getPoseParameters(self) -> np.ndarray:
        P = C.c_int32()
        self._check(self._lib.dmsa_get_params(self._ctx, None, C.byref(P)), "get_params")
        out = np.zeros(P.value)
        self._check(self._lib.dmsa_get_params(self._ctx, capi.ptr(out, C.c_double), C.byref(P)), "get_params")
        return out

    def setPoseParameters(self, params):
        params = np.ascontiguousarray(params, np.float64)
        self._check(self._lib.dmsa_set_params(self._ctx, capi.ptr(params, C.c_double)), "set_params")

    def getAdditionalErrorTerms(self) -> np.ndarray:
        """updateAdditionalErrors() + getAdditionalErrorTerms() for the current pose parameters of the resident problem."""
        out, n = np.zeros(4096), C.c_int32(0)
        self._check(self._lib.dmsa_additional_errors(self._ctx, capi.ptr(out, C.c_double), out.shape[0], C.byref(n)), "additional_errors")
        return out[: n.value].copy()

    def numTableRows(self) -> int:
        r = C.c_int32()
        self._check(self._lib.dmsa_num_table_rows(self._ctx, C.byref(r)), "num_table_rows")
        return r.value

    def poseTables(self, params, download: bool = True):
        params = np.ascontiguousarray(np.atleast_2d(params), np.float64)
        B = params.shape[0]
        out = np.zeros((B, self.numTableRows(), 12), np.float32) if download else None
        self._check(self._lib.dmsa_pose_tables(self._ctx, B, capi.ptr(params, C.c_double), capi.ptr(out, C.c_float)), "pose_tables")
        return out

    def setPoseTables(self, tables):
        tables = np.ascontiguousarray(tables, np.float32)
        if tables.ndim == 2:
            tables = tables[None]
        self._check(self._lib.dmsa_set_pose_tables(self._ctx, tables.shape[0], capi.ptr(tables, C.c_float)), "set_pose_tables")

    def updateGlobalPoints(self, b: int = 0, download: bool = True):
        out = np.zeros((self._num_points(), 4), np.float32) if download else None
        self._check(self._lib.dmsa_transform_points(self._ctx, int(b), capi.ptr(out, C.c_float)), "transform_points")
        return out

    def buildGaussians(self, settings: DmsaOptimSettings):
        cs = settings.to_c()
        M, Mm = C.c_int32(), C.c_int64()
        self._check(self._lib.dmsa_build_gaussians(self._ctx, C.byref(cs), C.byref(M), C.byref(Mm)), "build_gaussians")
        self._M, self._Mm = M.value, Mm.value
        return M.value, Mm.value

    def voxelLevel(self, level: int):
        n = self._num_points()
        info = capi.VoxelLevelInfo()
        code = np.zeros(n, np.uint64)
        key = np.zeros((n, 3), np.uint32)
        order = np.full(n, -1, np.int32)
        self._check(self._lib.dmsa_get_voxel_level(self._ctx, int(level), C.byref(info), capi.ptr(code, C.c_uint64), capi.ptr(key, C.c_uint32),
                                                   capi.ptr(order, C.c_int32)), "get_voxel_level")
        return info, code, key, order[: info.num_valid]

    def gaussians(self):
        M, Mm = self._M, self._Mm
        seg = np.zeros(M + 1, np.int32)
        memb = np.zeros(Mm, np.int32)
        info = np.zeros((M, 9), np.float32)
        w = np.zeros(M, np.float32)
        self._check(self._lib.dmsa_get_gaussians(self._ctx, capi.ptr(seg, C.c_int32), capi.ptr(memb, C.c_int32), capi.ptr(info, C.c_float),
                                                 capi.ptr(w, C.c_float)), "get_gaussians")
        return seg, memb, info, w

    def evalResiduals(self, B: int, download: bool = True):
        out = np.zeros((B, self._M)) if download else None
        self._check(self._lib.dmsa_eval_residuals(self._ctx, capi.ptr(out, C.c_double)), "eval_residuals")
        return out

    def normalEquations(self, P: int, h: float, lam: float, extra_rows=None):
        H = np.zeros((P, P))
        g = np.zeros(P)
        a = 0
        er = None
        if extra_rows is not None and np.size(extra_rows) > 0:
            er = np.ascontiguousarray(extra_rows, np.float64)
            a = er.shape[1]
        self._check(self._lib.dmsa_normal_equations(self._ctx, P, a, capi.ptr(er, C.c_double), float(h), float(lam), capi.ptr(H, C.c_double),
                                                    capi.ptr(g, C.c_double)), "normal_equations")
        return H, g  # H is symmetric, so col-major == row-major

    def detmathEval(self, fn: int, x, y=None) -> np.ndarray:
        """include/dmsa_detmath.h evaluated on the device (fn 0 sin, 1 cos, 2 acos, 3 atan2(y, x)) -- parity tests."""
        x = np.ascontiguousarray(x, np.float64)
        yy = None if y is None else np.ascontiguousarray(y, np.float64)
        out = np.zeros_like(x)
        self._check(self._lib.dmsa_detmath_eval(self._ctx, int(fn), capi.ptr(x, C.c_double), capi.ptr(yy, C.c_double) if yy is not None else None, x.size,
                                                capi.ptr(out, C.c_double)), "dmsa_detmath_eval")
        return out

    def sortPairs(self, keys, values, end_bit: int = 32):
        """The library's stable radix sort of (u32 key, u32 value) pairs on key bits [0, end_bit) -- test hook."""
        k = np.ascontiguousarray(keys, np.uint32)
        v = np.ascontiguousarray(values, np.uint32)
        ko, vo = np.zeros_like(k), np.zeros_like(v)
        self._check(self._lib.dmsa_sort_pairs(self._ctx, capi.ptr(k, C.c_uint32), capi.ptr(v, C.c_uint32), k.size, int(end_bit), capi.ptr(ko, C.c_uint32),
                                              capi.ptr(vo, C.c_uint32)), "dmsa_sort_pairs")
        return ko, vo

    def sortPairs64(self, keys, values, end_bit: int = 64):
        """The stable sort of (u64 key, u32 value) pairs that leaf codes wider than 32 bits go through (two 32-bit halves) -- test hook."""
        k = np.ascontiguousarray(keys, np.uint64)
        v = np.ascontiguousarray(values, np.uint32)
        ko, vo = np.zeros_like(k), np.zeros_like(v)
        self._check(self._lib.dmsa_sort_pairs64(self._ctx, capi.ptr(k, C.c_uint64), capi.ptr(v, C.c_uint32), k.size, int(end_bit), capi.ptr(ko, C.c_uint64),
                                                capi.ptr(vo, C.c_uint32)), "dmsa_sort_pairs64")
        return ko, vo

    def scan(self, values, inclusive: bool):
        """Prefix scan of an int32 array by the library's single-pass kernel -- test hook."""
        x = np.ascontiguousarray(values, np.int32)
        out = np.zeros_like(x)
        self._check(self._lib.dmsa_scan_i32(self._ctx, capi.ptr(x, C.c_int32), x.size, 1 if inclusive else 0, capi.ptr(out, C.c_int32)), "dmsa_scan_i32")
        return out

    def leafSegments(self, codes_sorted, code_bits: int):
        """Segmentation of sorted leaf codes by the single-pass kernel -- test hook.  Returns (leaf_of_pos, leaf_start[:num_leaves + 1])."""
        c = np.ascontiguousarray(codes_sorted, np.uint32)
        of_pos, start, nl = np.zeros(c.size, np.int32), np.zeros(c.size + 1, np.int32), C.c_int32(0)
        self._check(self._lib.dmsa_leaf_segments(self._ctx, capi.ptr(c, C.c_uint32), c.size, int(code_bits), capi.ptr(of_pos, C.c_int32), capi.ptr(start, C.c_int32),
                                                 C.byref(nl)), "dmsa_leaf_segments")
        return of_pos, start[:nl.value + 1]

    def lmSolveDevice(self, H_damped, g, alpha: float, max_step: float = float("inf")):
        """The LM step of DmsaOptimizer.h:107-128 as the device-resident loop computes it -- test hook.  Returns (step, has_nan)."""
        H = np.ascontiguousarray(H_damped, np.float64)
        gv = np.ascontiguousarray(g, np.float64)
        P = gv.size
        assert H.shape == (P, P)
        Hc = np.ascontiguousarray(H.T)  # column-major
        step, nan = np.zeros(P), C.c_int32(0)
        self._check(self._lib.dmsa_lm_solve_device(self._ctx, capi.ptr(Hc, C.c_double), capi.ptr(gv, C.c_double), P, float(alpha), float(max_step),
                                                   capi.ptr(step, C.c_double), C.byref(nan)), "dmsa_lm_solve_device")
        return step, bool(nan.value)

    def serialFallbackSums(self, reset: bool = False) -> int:
        """(Gaussian, sub-batch) double sums that were chained member by member because the exactness test of the parallel sum failed."""
        v = C.c_uint64(0)
        self._check(self._lib.dmsa_serial_fallback_sums(self._ctx, int(reset), C.byref(v)), "dmsa_serial_fallback_sums")
        return int(v.value)

    def timing(self, reset: bool = False) -> capi.Timing:
        t = capi.Timing()
        self._check(self._lib.dmsa_get_timing(self._ctx, C.byref(t), int(reset)), "get_timing")
        return t

    def trace(self, capacity: int = 256):
        buf = (capi.IterTrace * capacity)()
        n = self._lib.dmsa_get_trace(self._ctx, buf, capacity)
        return [dict(M=t.M, M1=t.M1, Mm=t.Mm, error0=t.error0, step_norm=t.step_norm, best_k=t.best_k) for t in buf[: max(n, 0)]]

    def synchronize(self):
        self._check(self._lib.dmsa_synchronize(self._ctx), "synchronize")
