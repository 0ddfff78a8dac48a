"""ctypes loader for the CPU oracle (oracle/libdmsa_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never by the dmsa_lidar_slam_amd package.  See oracle/dmsa_oracle.h for what the oracle is
(a CPU restatement of the reference's algorithm, parity unpinned by the reference itself).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from dmsa_lidar_slam_amd import _capi as capi
from dmsa_lidar_slam_amd.problems import ContinuousTrajectory, DmsaOptimSettings, MapManagement

_HERE = os.path.dirname(os.path.abspath(__file__))
# DMSA_ORACLE_LIB: time another build of the same source (scripts/cpu_variants.py: -O1 like the reference, -O3 -march=native)
LIB = os.environ.get("DMSA_ORACLE_LIB") or os.path.join(_HERE, "libdmsa_oracle.so")
_lib = None


class IterTrace(C.Structure):
    _fields_ = [("M", C.c_int32), ("M1", C.c_int32), ("Mm", C.c_int64), ("error0", C.c_double), ("step_norm", C.c_double),
                ("best_k", C.c_int32), ("pad", C.c_int32)]


def build() -> None:
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    L = C.CDLL(LIB)
    dp, fp, ip, lp = capi.c_double_p, capi.c_float_p, capi.c_int32_p, capi.c_int64_p
    vp = C.c_void_p
    L.orc_axang2rotm.argtypes = [dp, dp]
    L.orc_rotm2axang.argtypes = [dp, dp]
    L.orc_slerp.argtypes = [dp, dp, C.c_double, dp]
    L.orc_relative2global.argtypes = [C.c_int, dp, dp, dp, dp]
    L.orc_global2relative.argtypes = [C.c_int, dp, dp, dp, dp]
    L.orc_barycentric_rational.argtypes = [dp, dp, C.c_int, C.c_int, dp, C.c_int, dp]
    L.orc_window_pose_table.argtypes = [C.POINTER(capi.WindowProblem), fp, dp]
    L.orc_keyframe_pose_table.argtypes = [C.POINTER(capi.KeyframeProblem), fp]
    L.orc_transform_points.argtypes = [fp, fp, ip, C.c_int64, fp]
    L.orc_voxelize.argtypes = [fp, C.c_int64, C.c_double, C.POINTER(capi.VoxelLevelInfo), capi.c_uint64_p, capi.c_uint32_p, ip]
    L.orc_build_gaussians.argtypes = [fp, fp, ip, C.c_int64, C.c_float, C.POINTER(capi.Settings)]
    L.orc_build_gaussians.restype = vp
    L.orc_gaussians_free.argtypes = [vp]
    L.orc_gaussians_count.argtypes = [vp]
    L.orc_gaussians_count_level1.argtypes = [vp]
    L.orc_gaussians_memberships.argtypes = [vp]
    L.orc_gaussians_memberships.restype = C.c_int64
    L.orc_gaussians_get.argtypes = [vp, ip, ip, fp, fp]
    L.orc_gaussians_set_info.argtypes = [vp, fp, fp]
    L.orc_gaussians_get_fit.argtypes = [vp, fp, fp, fp]
    L.orc_eval_residuals.argtypes = [vp, fp, dp]
    L.orc_eigen_mean_f32.argtypes = [fp, C.c_int64, C.c_int64]
    L.orc_eigen_mean_f32.restype = C.c_float
    L.orc_optimize_window.argtypes = [C.POINTER(capi.WindowProblem), C.POINTER(capi.Settings), C.POINTER(capi.Report), fp,
                                      C.POINTER(IterTrace), C.c_int32, C.c_int32]
    L.orc_optimize_keyframes.argtypes = [C.POINTER(capi.KeyframeProblem), C.POINTER(capi.Settings), C.POINTER(capi.Report), fp,
                                         C.POINTER(IterTrace), C.c_int32, C.c_int32]
    L.orc_window_additional_errors.argtypes = [C.POINTER(capi.WindowProblem), dp]
    L.orc_keyframe_additional_errors.argtypes = [C.POINTER(capi.KeyframeProblem), dp]
    L.orc_lm_step.argtypes = [dp, dp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, dp, dp, dp]
    L.orc_lm_step_from_jacobian.argtypes = [dp, dp, C.c_int32, C.c_int32, C.c_double, C.c_double, dp, dp, dp]
    L.orc_stage_dump_window.argtypes = [C.POINTER(capi.WindowProblem), C.POINTER(capi.Settings), fp, fp, C.c_int32, C.c_char_p]
    L.orc_stage_dump_keyframes.argtypes = [C.POINTER(capi.KeyframeProblem), C.POINTER(capi.Settings), fp, fp, C.c_int32, C.c_char_p]
    _lib = L
    return L


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def axang2rotm(w):
    w = _d(w)
    out = np.zeros(9)
    lib().orc_axang2rotm(capi.ptr(w, C.c_double), capi.ptr(out, C.c_double))
    return out.reshape(3, 3).T.copy()


def rotm2axang(R):
    Rc = _d(np.asarray(R).T)
    out = np.zeros(3)
    lib().orc_rotm2axang(capi.ptr(Rc, C.c_double), capi.ptr(out, C.c_double))
    return out


def slerp(a, b, t):
    a, b = _d(a), _d(b)
    out = np.zeros(3)
    lib().orc_slerp(capi.ptr(a, C.c_double), capi.ptr(b, C.c_double), float(t), capi.ptr(out, C.c_double))
    return out


def relative2global(ro, rt):
    ro, rt = _d(ro), _d(rt)
    go, gt = np.zeros_like(ro), np.zeros_like(rt)
    lib().orc_relative2global(ro.shape[0], capi.ptr(ro, C.c_double), capi.ptr(rt, C.c_double), capi.ptr(go, C.c_double), capi.ptr(gt, C.c_double))
    return go, gt


def global2relative(go, gt):
    go, gt = _d(go), _d(gt)
    ro, rt = np.zeros_like(go), np.zeros_like(gt)
    lib().orc_global2relative(go.shape[0], capi.ptr(go, C.c_double), capi.ptr(gt, C.c_double), capi.ptr(ro, C.c_double), capi.ptr(rt, C.c_double))
    return ro, rt


def barycentric_rational(x, y, t, d=2):
    x, y, t = _d(x), _d(y), _d(t)
    out = np.zeros_like(t)
    rc = lib().orc_barycentric_rational(capi.ptr(x, C.c_double), capi.ptr(y, C.c_double), len(x), d, capi.ptr(t, C.c_double), len(t), capi.ptr(out, C.c_double))
    if rc != 0:
        raise ValueError("coincident nodes")
    return out


def detmath_eval(fn: int, x, y=None):
    """include/dmsa_detmath.h compiled for the host: fn 0 sin, 1 cos, 2 acos, 3 atan2(y, x)."""
    x = np.ascontiguousarray(x, np.float64)
    y = np.zeros_like(x) if y is None else np.ascontiguousarray(y, np.float64)
    out = np.zeros_like(x)
    lib().orc_detmath_eval.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_long, C.POINTER(C.c_double)]
    lib().orc_detmath_eval(int(fn), capi.ptr(x, C.c_double), capi.ptr(y, C.c_double), x.size, capi.ptr(out, C.c_double))
    return out


def window_pose_table(prob: ContinuousTrajectory):
    n_t = prob.trajTime.shape[0]
    table = np.zeros((n_t, 12), np.float32)
    dense = np.zeros((n_t, 3))
    cp = prob.to_c()
    lib().orc_window_pose_table(C.byref(cp), capi.ptr(table, C.c_float), capi.ptr(dense, C.c_double))
    return table, dense


def keyframe_pose_table(prob: MapManagement):
    table = np.zeros((prob.numFrames, 12), np.float32)
    cp = prob.to_c()
    lib().orc_keyframe_pose_table(C.byref(cp), capi.ptr(table, C.c_float))
    return table


def transform_points(table, xyz4, rows):
    table = np.ascontiguousarray(table, np.float32)
    xyz4 = np.ascontiguousarray(xyz4, np.float32)
    rows = np.ascontiguousarray(rows, np.int32)
    out = np.zeros_like(xyz4)
    lib().orc_transform_points(capi.ptr(table, C.c_float), capi.ptr(xyz4, C.c_float), capi.ptr(rows, C.c_int32), xyz4.shape[0], capi.ptr(out, C.c_float))
    return out


def voxelize(xyz4, resolution: float):
    xyz4 = np.ascontiguousarray(xyz4, np.float32)
    n = xyz4.shape[0]
    info = capi.VoxelLevelInfo()
    code = np.zeros(n, np.uint64)
    key = np.zeros((n, 3), np.uint32)
    order = np.full(n, -1, np.int32)
    rc = lib().orc_voxelize(capi.ptr(xyz4, C.c_float), n, float(resolution), C.byref(info), capi.ptr(code, C.c_uint64), capi.ptr(key, C.c_uint32), capi.ptr(order, C.c_int32))
    if rc != 0:
        raise RuntimeError(f"orc_voxelize rc={rc}")
    return info, code, key, order[: info.num_valid]


class Gaussians:
    """Handle on the oracle's Gaussians for stage-level parity tests."""

    def __init__(self, xyz4, ids, min_grid_size: float, settings: DmsaOptimSettings, normals4=None):
        self._xyz = np.ascontiguousarray(xyz4, np.float32)
        self._ids = np.ascontiguousarray(ids, np.int32)
        self._nrm = None if normals4 is None else np.ascontiguousarray(normals4, np.float32)
        cs = settings.to_c()
        self._h = lib().orc_build_gaussians(capi.ptr(self._xyz, C.c_float), capi.ptr(self._nrm, C.c_float), capi.ptr(self._ids, C.c_int32),
                                            self._xyz.shape[0], float(min_grid_size), C.byref(cs))
        self.M = lib().orc_gaussians_count(self._h)
        self.M1 = lib().orc_gaussians_count_level1(self._h)
        self.Mm = lib().orc_gaussians_memberships(self._h)
        self.seg_offset = np.zeros(self.M + 1, np.int32)
        self.members = np.zeros(self.Mm, np.int32)
        self.info = np.zeros((self.M, 9), np.float32)
        self.weights = np.zeros(self.M, np.float32)
        lib().orc_gaussians_get(self._h, capi.ptr(self.seg_offset, C.c_int32), capi.ptr(self.members, C.c_int32), capi.ptr(self.info, C.c_float), capi.ptr(self.weights, C.c_float))

    def fit_sums(self):
        """(mean M x 3, covariance before limitCovariance M x 9 column-major, pow(-1) of the member counts M)"""
        mean, cov, raw = np.zeros((self.M, 3), np.float32), np.zeros((self.M, 9), np.float32), np.zeros(self.M, np.float32)
        lib().orc_gaussians_get_fit(self._h, capi.ptr(mean, C.c_float), capi.ptr(cov, C.c_float), capi.ptr(raw, C.c_float))
        return mean, cov, raw

    def set_info(self, info, weights):
        info = np.ascontiguousarray(info, np.float32)
        weights = np.ascontiguousarray(weights, np.float32)
        lib().orc_gaussians_set_info(self._h, capi.ptr(info, C.c_float), capi.ptr(weights, C.c_float))
        self.info, self.weights = info.reshape(self.M, 9), weights

    def residuals(self, xyz4_global):
        x = np.ascontiguousarray(xyz4_global, np.float32)
        e = np.zeros(self.M)
        lib().orc_eval_residuals(self._h, capi.ptr(x, C.c_float), capi.ptr(e, C.c_double))
        return e

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_gaussians_free(self._h)
            self._h = None


def eigen_mean_f32(x, offset_floats=0):
    """DenseBase::mean() of a contiguous float vector in Eigen 3.4's linear vectorised redux order (SSE2 packets); `offset_floats`: where the
    vector starts behind a 16-byte boundary (column c of an n x 3 column-major matrix: c * n)."""
    x = np.ascontiguousarray(x, np.float32)
    return np.float32(lib().orc_eigen_mean_f32(x.ctypes.data_as(capi.c_float_p), x.size, int(offset_floats)))


def set_eigen_l1_bytes(nbytes: int) -> None:
    """L1 data cache size of the machine the reference runs on (Eigen sizes the depth blocks of centered^T * centered from it)."""
    L = lib()
    L.orc_set_eigen_l1_bytes.argtypes = [C.c_int]
    L.orc_set_eigen_l1_bytes(int(nbytes))


def eigen_gemm_kc(depth: int) -> int:
    L = lib()
    L.orc_eigen_gemm_kc.argtypes, L.orc_eigen_gemm_kc.restype = [C.c_int64], C.c_int64
    return int(L.orc_eigen_gemm_kc(int(depth)))


def eigen_gemm_dot_f32(a, b):
    """sum_k a[k] * b[k] in the float order of Eigen 3.4's (3 x n) * (n x 3) product (Gaussians.h:147): lazy product below 14, gebp chains in depth blocks above."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    L = lib()
    L.orc_eigen_gemm_dot_f32.argtypes, L.orc_eigen_gemm_dot_f32.restype = [capi.c_float_p, capi.c_float_p, C.c_int64], C.c_float
    return np.float32(L.orc_eigen_gemm_dot_f32(a.ctypes.data_as(capi.c_float_p), b.ctypes.data_as(capi.c_float_p), a.size))


def eigensolver3f(A):
    """EigenSolver<Matrix3f> (Eigen 3.4.0, oracle/eigensolver3f.h) on an (n, 3, 3) array of matrices: eigenvalues (re, im) in the order of T's
    diagonal, real parts of the normalised eigenvectors (n, 3, 3) [row, col], Francis QR iterations, info, complex pairs."""
    A = np.ascontiguousarray(np.asarray(A, np.float32).reshape(-1, 3, 3))
    n = A.shape[0]
    At = np.ascontiguousarray(A.transpose(0, 2, 1))  # column-major per matrix
    re, im = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    Vt = np.zeros((n, 3, 3), np.float32)
    it, info, pairs = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    L = lib()
    L.orc_eigensolver3f.argtypes = [capi.c_float_p, C.c_int64, capi.c_float_p, capi.c_float_p, capi.c_float_p, capi.c_int32_p, capi.c_int32_p, capi.c_int32_p]
    L.orc_eigensolver3f.restype = None
    L.orc_eigensolver3f(At.ctypes.data_as(capi.c_float_p), n, re.ctypes.data_as(capi.c_float_p), im.ctypes.data_as(capi.c_float_p),
                        Vt.ctypes.data_as(capi.c_float_p), it.ctypes.data_as(capi.c_int32_p), info.ctypes.data_as(capi.c_int32_p),
                        pairs.ctypes.data_as(capi.c_int32_p))
    return re, im, np.ascontiguousarray(Vt.transpose(0, 2, 1)), it, info, pairs


def limit_covariance(cov):
    """Gaussians::limitCovariance (Gaussians.h:181-201) on an (n, 3, 3) array [row, col]."""
    A = np.ascontiguousarray(np.asarray(cov, np.float32).reshape(-1, 3, 3))
    At = np.ascontiguousarray(A.transpose(0, 2, 1))
    out = np.zeros_like(At)
    L = lib()
    L.orc_limit_covariance.argtypes, L.orc_limit_covariance.restype = [capi.c_float_p, C.c_int64, capi.c_float_p], None
    L.orc_limit_covariance(At.ctypes.data_as(capi.c_float_p), A.shape[0], out.ctypes.data_as(capi.c_float_p))
    return np.ascontiguousarray(out.transpose(0, 2, 1))


def info_from_covariance(cov9):
    """limitCovariance + inverse (Gaussians.h:150-154) on an (M, 9) array of column-major covariances -> (M, 9) column-major information matrices."""
    c = np.ascontiguousarray(cov9, np.float32).reshape(-1, 9)
    out = np.zeros_like(c)
    L = lib()
    L.orc_info_from_covariance.argtypes, L.orc_info_from_covariance.restype = [capi.c_float_p, C.c_int64, capi.c_float_p], None
    L.orc_info_from_covariance(c.ctypes.data_as(capi.c_float_p), c.shape[0], out.ctypes.data_as(capi.c_float_p))
    return out


def limitcov_eigenpairs(cov9):
    """(eigenvalues M x 3, eigenvectors M x 9 column-major) that limitCovariance decomposes an (M, 9) array of column-major covariances with."""
    c = np.ascontiguousarray(cov9, np.float32).reshape(-1, 9)
    ev, V = np.zeros((c.shape[0], 3), np.float32), np.zeros_like(c)
    L = lib()
    L.orc_limitcov_eigenpairs.argtypes, L.orc_limitcov_eigenpairs.restype = [capi.c_float_p, C.c_int64, capi.c_float_p, capi.c_float_p], None
    L.orc_limitcov_eigenpairs(c.ctypes.data_as(capi.c_float_p), c.shape[0], ev.ctypes.data_as(capi.c_float_p), V.ctypes.data_as(capi.c_float_p))
    return ev, V


def limitcov_stats(reset=False):
    """(calls, QR iterations, max iterations of one call, complex pairs, not converged) of limitCovariance since load / the last reset."""
    out = (C.c_int64 * 5)()
    L = lib()
    L.orc_limitcov_stats.argtypes, L.orc_limitcov_stats.restype = [C.POINTER(C.c_int64), C.c_int], None
    L.orc_limitcov_stats(out, 1 if reset else 0)
    return dict(calls=out[0], qr_iterations=out[1], max_iterations=out[2], complex_pairs=out[3], not_converged=out[4])


def _run(fn, prob, settings, fixed_iters, want_global, npts):
    cp = prob.to_c()
    cs = settings.to_c()
    rep = capi.Report()
    trace = (IterTrace * max(1, settings.num_iter))()
    gl = np.zeros((npts, 4), np.float32) if want_global else None
    rc = fn(C.byref(cp), C.byref(cs), C.byref(rep), capi.ptr(gl, C.c_float), trace, settings.num_iter, int(fixed_iters))
    if rc != 0:
        raise RuntimeError(f"oracle optimize rc={rc}")
    tr = [dict(M=t.M, M1=t.M1, Mm=t.Mm, error0=t.error0, step_norm=t.step_norm, best_k=t.best_k) for t in trace[: rep.iterations]]
    return rep, gl, tr


def _static_sigs(L):
    fp, ip, lp = capi.c_float_p, capi.c_int32_p, capi.c_int64_p
    u8p = C.POINTER(C.c_uint8)
    L.orc_radius_exists.argtypes = [fp, C.c_int64, fp, C.c_int64, C.c_float, u8p, C.c_int]
    L.orc_select_static_points.argtypes = [C.POINTER(capi.StaticSelectProblem), fp, ip, C.c_int64, ip, C.POINTER(capi.StaticSelectResult)]
    L.orc_get_overlap.argtypes = [fp, C.c_int64, fp, C.c_int64, C.c_float, fp, lp]
    L.orc_glibc_rand.argtypes = [C.c_uint32, C.c_int32, ip]
    L.orc_glibc_rand.restype = None
    L.orc_random_grid_downsampling.argtypes = [fp, C.c_int64, C.c_float, C.c_uint32, ip, C.c_int64, lp]
    L.orc_preprocess_scan.argtypes = [fp, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_uint32, fp, fp, ip, C.c_int64, lp, fp]
    return L


def radius_exists(cloud, query, radius, brute=False):
    """Is there a cloud point within `radius` of every query (squared L2_Simple float distance <= radius^2)?"""
    L = _static_sigs(lib())
    a = np.ascontiguousarray(cloud, np.float32)
    q = np.ascontiguousarray(query, np.float32)
    out = np.zeros(max(1, q.shape[0]), np.uint8)
    rc = L.orc_radius_exists(capi.ptr(a, C.c_float), a.shape[0], capi.ptr(q, C.c_float), q.shape[0], float(np.float32(radius)),
                             out.ctypes.data_as(C.POINTER(C.c_uint8)), int(brute))
    if rc != 0:
        raise RuntimeError(f"orc_radius_exists rc={rc}")
    return out[: q.shape[0]].astype(bool)


def select_static_points(prob):
    """DmsaSlam::addStaticPoints keyframe loop (DmsaSlam.h:300-344) on a static_points.StaticSelectProblem."""
    from dmsa_lidar_slam_amd.static_points import StaticSelection

    L = _static_sigs(lib())
    cp = prob.to_c()
    cap = prob.keyPoints.shape[0]
    xyz, ids = np.zeros((max(cap, 1), 4), np.float32), np.zeros(max(cap, 1), np.int32)
    ov = np.zeros(max(1, prob.keyframeIds.shape[0]), np.int32)
    res = capi.StaticSelectResult()
    rc = L.orc_select_static_points(C.byref(cp), capi.ptr(xyz, C.c_float), capi.ptr(ids, C.c_int32), cap, capi.ptr(ov, C.c_int32), C.byref(res))
    if rc != 0:
        raise RuntimeError(f"orc_select_static_points rc={rc}")
    m = res.num_static
    return StaticSelection(xyz[:m].copy(), ids[:m].copy(), ov[: prob.keyframeIds.shape[0]].copy(), res.keyframe_id, res.min_related_key_id, res.max_overlap)


def get_overlap(pc1, pc2, max_dist):
    L = _static_sigs(lib())
    a, b = np.ascontiguousarray(pc1, np.float32), np.ascontiguousarray(pc2, np.float32)
    ov, nc = C.c_float(0.0), C.c_int64(0)
    rc = L.orc_get_overlap(capi.ptr(a, C.c_float), a.shape[0], capi.ptr(b, C.c_float), b.shape[0], float(np.float32(max_dist)), C.byref(ov), C.byref(nc))
    if rc != 0:
        raise RuntimeError(f"orc_get_overlap rc={rc}")
    return float(ov.value), int(nc.value)


def glibc_rand(seed, count):
    L = _static_sigs(lib())
    out = np.zeros(count, np.int32)
    L.orc_glibc_rand(int(seed) & 0xFFFFFFFF, count, capi.ptr(out, C.c_int32))
    return out


def random_grid_downsampling(xyz, grid_size, seed):
    L = _static_sigs(lib())
    a = np.ascontiguousarray(xyz, np.float32)
    out = np.zeros(max(1, a.shape[0]), np.int32)
    n = C.c_int64(0)
    rc = L.orc_random_grid_downsampling(capi.ptr(a, C.c_float), a.shape[0], float(np.float32(grid_size)), int(seed) & 0xFFFFFFFF, capi.ptr(out, C.c_int32),
                                        a.shape[0], C.byref(n))
    if rc != 0:
        raise RuntimeError(f"orc_random_grid_downsampling rc={rc}")
    return out[: n.value].copy()


def preprocess_scan(raw, seed, max_num_points_per_scan=3000, min_dist_ds=30.0, min_dist=0.0, lidar_to_imu=None):
    """DmsaSlam::preProcess (DmsaSlam.h:569-634): (filtered xyz1 in the IMU frame, index into raw, gridSize)."""
    L = _static_sigs(lib())
    a = np.ascontiguousarray(raw, np.float32)
    T = np.eye(4, dtype=np.float32) if lidar_to_imu is None else np.asarray(lidar_to_imu, np.float32).reshape(4, 4)
    tf = np.ascontiguousarray(T.T.reshape(-1))  # Eigen storage: column-major
    cap = a.shape[0]
    xyz, src = np.zeros((max(cap, 1), 4), np.float32), np.zeros(max(cap, 1), np.int32)
    n, grid = C.c_int64(0), C.c_float(0.0)
    rc = L.orc_preprocess_scan(capi.ptr(a, C.c_float), a.shape[0], int(max_num_points_per_scan), float(np.float32(min_dist_ds)), float(np.float32(min_dist)),
                               int(seed) & 0xFFFFFFFF, capi.ptr(tf, C.c_float), capi.ptr(xyz, C.c_float), capi.ptr(src, C.c_int32), cap, C.byref(n), C.byref(grid))
    if rc != 0:
        raise RuntimeError(f"orc_preprocess_scan rc={rc}")
    return xyz[: n.value].copy(), src[: n.value].copy(), float(grid.value)


def set_threads(n: int) -> None:
    """Evaluation-parallel variant of the CPU baseline (OpenMP over the forward differences / line-search trials; without IMU rows
    results are bit-identical to one thread).  1 = the reference's own execution order."""
    L = lib()
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_set_threads(int(n))


def optimize_window(prob: ContinuousTrajectory, settings: DmsaOptimSettings, fixed_iters=False, want_global=False):
    """Runs optimizeSet on `prob` IN PLACE (relOrientations / relTranslations are updated)."""
    n = prob.localPoints.shape[0] + prob.staticPoints.shape[0]
    return _run(lib().orc_optimize_window, prob, settings, fixed_iters, want_global, n)


def optimize_keyframes(prob: MapManagement, settings: DmsaOptimSettings, fixed_iters=False, want_global=False):
    return _run(lib().orc_optimize_keyframes, prob, settings, fixed_iters, want_global, prob.localPoints.shape[0])


def window_additional_errors(prob: ContinuousTrajectory):
    out = np.zeros(max(1, prob.numControlPoses))
    cp = prob.to_c()
    n = lib().orc_window_additional_errors(C.byref(cp), capi.ptr(out, C.c_double))
    return out[:n]


def keyframe_additional_errors(prob: MapManagement):
    out = np.zeros(2 * prob.numFrames + 2)
    cp = prob.to_c()
    n = lib().orc_keyframe_additional_errors(C.byref(cp), capi.ptr(out, C.c_double))
    return out[:n]


def stage_dump(prob, settings: DmsaOptimSettings, path: str, inject_info=None, inject_weights=None):
    """Iteration 0 of optimizeSet stage by stage into a 'DMSAST03' file (dmsa_lidar_slam_amd/dump.py: read_stage_dump); inject_info (M x 9) /
    inject_weights (M) replace the fitted information matrices / weights before the residuals are evaluated.  `prob` is not modified."""
    q = prob.copy()  # (kept alive: the C struct points into its arrays)
    cp = q.to_c()
    cs = settings.to_c()
    ii = np.ascontiguousarray(inject_info, np.float32) if inject_info is not None else None
    iw = np.ascontiguousarray(inject_weights, np.float32) if inject_weights is not None else None
    m = ii.shape[0] if ii is not None else (iw.shape[0] if iw is not None else 0)
    fn = lib().orc_stage_dump_window if isinstance(prob, ContinuousTrajectory) else lib().orc_stage_dump_keyframes
    rc = fn(C.byref(cp), C.byref(cs), ii.ctypes.data_as(capi.c_float_p) if ii is not None else None, iw.ctypes.data_as(capi.c_float_p) if iw is not None else None,
            int(m), path.encode())
    if rc != capi.DMSA_OK:
        raise RuntimeError(f"orc_stage_dump failed with {rc}")
    from dmsa_lidar_slam_amd import dump

    return dump.read_stage_dump(path)


def lm_step_from_jacobian(e0, J, lam, alpha):
    """H = J^T J + lambda I (damped), g = J^T e0 and the step of DmsaOptimizer.h:107-113 from a given Jacobian (rows x P)."""
    e0 = _d(e0)
    J = np.asarray(J, np.float64)
    rows, P = J.shape
    Jc = np.ascontiguousarray(J.T)  # column-major rows x P
    H, g, step = np.zeros((P, P)), np.zeros(P), np.zeros(P)
    lib().orc_lm_step_from_jacobian(capi.ptr(e0, C.c_double), capi.ptr(Jc, C.c_double), rows, P, float(lam), float(alpha), capi.ptr(H, C.c_double),
                                    capi.ptr(g, C.c_double), capi.ptr(step, C.c_double))
    return H.T.copy(), g, step


def lm_step(e0, e_batch, h, lam, alpha):
    e0 = _d(e0)
    eb = _d(e_batch)
    P, rows = eb.shape
    H = np.zeros((P, P))
    g = np.zeros(P)
    step = np.zeros(P)
    lib().orc_lm_step(capi.ptr(e0, C.c_double), capi.ptr(eb, C.c_double), rows, P, float(h), float(lam), float(alpha),
                      capi.ptr(H, C.c_double), capi.ptr(g, C.c_double), capi.ptr(step, C.c_double))
    return H, g, step


# ---- the oracle's window setup, same Python surface as dmsa_lidar_slam_amd.window_setup ------------------------
def _setup_sigs(L):
    dp, ip, vp = capi.c_double_p, capi.c_int32_p, C.c_void_p
    L.orc_imu_buffer_create.argtypes, L.orc_imu_buffer_create.restype = [C.c_int32], vp
    L.orc_imu_buffer_destroy.argtypes, L.orc_imu_buffer_destroy.restype = [vp], None
    L.orc_imu_buffer_add.argtypes, L.orc_imu_buffer_add.restype = [vp, dp, dp, C.c_double], None
    L.orc_imu_buffer_closest.argtypes = [vp, C.c_double, dp, dp, dp]
    L.orc_imu_buffer_state.argtypes, L.orc_imu_buffer_state.restype = [vp, ip, ip, dp], None
    L.orc_traj_dims.argtypes = [C.c_double, C.c_double, C.c_double, dp, ip]
    L.orc_traj_grids.argtypes = [C.c_double, C.c_double, C.c_int32, C.c_int32, dp, dp, ip]
    L.orc_traj_tform_indices.argtypes = [dp, C.c_int64, C.c_double, dp, C.c_int32, ip]
    L.orc_traj_transfer_imu.argtypes = [vp, C.c_double, dp, C.c_int32, dp, dp, dp]
    L.orc_traj_preint_factors.argtypes = [C.c_int32, C.c_int32, ip, C.c_double, dp, dp, dp, dp, dp, dp, dp, dp, dp]
    L.orc_traj_update_initial_guess.argtypes = [ip, C.POINTER(capi.TrajState), C.POINTER(capi.TrajState), C.c_int32]
    return L


class ImuBuffer:
    """ImuBuffer.h:14-175 (oracle)."""

    def __init__(self, maxNumMeas: int = 10000):
        self._L = _setup_sigs(lib())
        self._h = self._L.orc_imu_buffer_create(int(maxNumMeas))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_imu_buffer_destroy(self._h)
            self._h = None

    def addMeasurement(self, AccVec, AngVelVec, stamp):
        a, w = np.ascontiguousarray(AccVec, np.float64), np.ascontiguousarray(AngVelVec, np.float64)
        self._L.orc_imu_buffer_add(self._h, capi.ptr(a, C.c_double), capi.ptr(w, C.c_double), float(stamp))

    def getClosestMeasurement(self, t):
        a, w, d = np.zeros(3), np.zeros(3), C.c_double(0.0)
        rc = self._L.orc_imu_buffer_closest(self._h, float(t), capi.ptr(a, C.c_double), capi.ptr(w, C.c_double), C.byref(d))
        if rc != 0:
            raise RuntimeError(f"orc_imu_buffer_closest rc={rc}")
        return a, w, d.value

    def state(self):
        n, o, b = C.c_int32(0), C.c_int32(0), np.zeros(3)
        self._L.orc_imu_buffer_state(self._h, C.byref(n), C.byref(o), capi.ptr(b, C.c_double))
        return n.value, o.value, b


class WindowSetup:
    """The setup half of ContinuousTrajectory (ContinuousTrajectory.h:228-568) on window_setup.TrajectoryState objects (oracle)."""

    def __init__(self):
        self._L = _setup_sigs(lib())

    def initTraj(self, t_min, t_max, numControlPoses, useImu, dtResIn):
        from dmsa_lidar_slam_amd.window_setup import new_state

        hor, n = C.c_double(0.0), C.c_int32(0)
        self._L.orc_traj_dims(float(t_min), float(t_max), float(dtResIn), C.byref(hor), C.byref(n))
        tt, st, pi = np.zeros(n.value), np.zeros(int(numControlPoses)), np.zeros(int(numControlPoses), np.int32)
        self._L.orc_traj_grids(hor.value, float(dtResIn), n.value, int(numControlPoses), capi.ptr(tt, C.c_double), capi.ptr(st, C.c_double), capi.ptr(pi, C.c_int32))
        return new_state(t_min, hor.value, dtResIn, n.value, st, tt, pi, useImu)

    def transferImuMeasurements(self, traj, imuBuffer):
        traj.accMeas, traj.angVelMeas = np.zeros((traj.n_total, 3)), np.zeros((traj.n_total, 3))
        worst = C.c_double(0.0)
        rc = self._L.orc_traj_transfer_imu(imuBuffer._h, traj.t0, capi.ptr(traj.trajTime, C.c_double), traj.n_total, capi.ptr(traj.accMeas, C.c_double),
                                           capi.ptr(traj.angVelMeas, C.c_double), C.byref(worst))
        if rc != 0:
            raise RuntimeError(f"orc_traj_transfer_imu rc={rc}")
        return worst.value

    def updatePreintFactors(self, traj, gyr_cov, acc_cov):
        c = traj.numControlPoses
        g, a = np.ascontiguousarray(np.asarray(gyr_cov, np.float64).T), np.ascontiguousarray(np.asarray(acc_cov, np.float64).T)
        rot, pos, vel, cov, hor = np.zeros((c, 3, 3)), np.zeros((c, 3)), np.zeros((c, 3)), np.zeros((c, 9, 9)), np.zeros(3)
        self._L.orc_traj_preint_factors(traj.n_total, c, capi.ptr(traj.paramIndices, C.c_int32), traj.dt_res, capi.ptr(traj.accMeas, C.c_double),
                                        capi.ptr(traj.angVelMeas, C.c_double), capi.ptr(g, C.c_double), capi.ptr(a, C.c_double), capi.ptr(rot, C.c_double),
                                        capi.ptr(pos, C.c_double), capi.ptr(vel, C.c_double), capi.ptr(cov, C.c_double), capi.ptr(hor, C.c_double))
        traj.preintImuRots = np.ascontiguousarray(np.transpose(rot, (0, 2, 1)))
        traj.CovPVRot_inv = np.ascontiguousarray(np.transpose(cov, (0, 2, 1)))
        traj.preintRelPositions, traj.preintRelVelocity, traj.preintPosComplHor = pos, vel, hor

    def updateInitialGuess(self, isInitialized, traj, oldTraj, useImu):
        flag = C.c_int32(int(bool(isInitialized)))
        cur = traj.to_c()
        old = oldTraj.to_c() if oldTraj is not None else None
        rc = self._L.orc_traj_update_initial_guess(C.byref(flag), C.byref(cur), C.byref(old) if old is not None else None, int(bool(useImu)))
        if rc != 0:
            raise RuntimeError(f"orc_traj_update_initial_guess rc={rc}")
        return bool(flag.value)

    def getSubmapGravityEstimate(self, traj):
        self._L.orc_traj_submap_gravity_estimate.argtypes = [C.POINTER(capi.TrajState), capi.c_double_p, capi.c_double_p]
        out = np.zeros(3)
        cs = traj.to_c()
        self._L.orc_traj_submap_gravity_estimate(C.byref(cs), capi.ptr(traj.preintPosComplHor, C.c_double), capi.ptr(out, C.c_double))
        return out

    def tformIdPerPoint(self, traj, pointStamps):
        st = np.ascontiguousarray(pointStamps, np.float64)
        out = np.zeros(max(1, st.shape[0]), np.int32)
        self._L.orc_traj_tform_indices(capi.ptr(st, C.c_double), st.shape[0], traj.t0, capi.ptr(traj.trajTime, C.c_double), traj.n_total, capi.ptr(out, C.c_int32))
        return out[: st.shape[0]]


# ---- wire formats -------------------------------------------------------------------------------------------------
def decode_pointcloud2(msg, sensor: str, delta_t_pcs: float = 0.0):
    """callbackPointCloud (dmsa_slam_ros.cpp:399-486) on a wire_formats.PointCloud2Msg."""
    L = lib()
    L.orc_decode_pointcloud2.argtypes = [C.POINTER(capi.PointCloud2), C.c_int32, capi.c_float_p, capi.c_double_p, capi.c_int32_p]
    n = int(msg.height) * int(msg.width)
    xyz, st, ids = np.zeros((max(n, 1), 4), np.float32), np.zeros(max(n, 1)), np.zeros(max(n, 1), np.int32)
    cm = msg.to_c(delta_t_pcs)
    L.orc_decode_pointcloud2(C.byref(cm), capi.SENSORS[sensor], capi.ptr(xyz, C.c_float), capi.ptr(st, C.c_double), capi.ptr(ids, C.c_int32))
    return xyz[:n], st[:n], ids[:n]


def format_tum_pose(stamp, pos, orient) -> str:
    L = lib()
    L.orc_format_tum_pose.argtypes = [C.c_double, capi.c_double_p, capi.c_double_p, C.c_char_p, C.c_int32]
    p, o = np.ascontiguousarray(pos, np.float64), np.ascontiguousarray(orient, np.float64)
    buf = C.create_string_buffer(512)
    n = L.orc_format_tum_pose(float(stamp), capi.ptr(p, C.c_double), capi.ptr(o, C.c_double), buf, 512)
    if n < 0:
        raise RuntimeError(f"orc_format_tum_pose rc={n}")
    return buf.raw[:n].decode()


def compose_nonkeyframe_pose(key_pos, key_orient, transl, orient):
    L = lib()
    L.orc_compose_nonkeyframe_pose.argtypes = [capi.c_double_p] * 6
    a = [np.ascontiguousarray(v, np.float64) for v in (key_pos, key_orient, transl, orient)]
    gp, go = np.zeros(3), np.zeros(3)
    L.orc_compose_nonkeyframe_pose(*[capi.ptr(v, C.c_double) for v in a], capi.ptr(gp, C.c_double), capi.ptr(go, C.c_double))
    return gp, go


# ---- keyframe creation --------------------------------------------------------------------------------------------
def update_normals(cloud, k=6, origin=(0.0, 0.0, 0.0), neighbours=False):
    """DmsaSlam::updateNormals (DmsaSlam.h:553-567): exhaustive neighbour search + pcl::NormalEstimation restated."""
    L = lib()
    L.orc_update_normals.argtypes = [capi.c_float_p, C.c_int64, C.c_int32, capi.c_float_p, capi.c_float_p, capi.c_int32_p]
    a = np.ascontiguousarray(cloud, np.float32)
    n = a.shape[0]
    vp = np.ascontiguousarray(origin, np.float32)
    out = np.zeros((max(n, 1), 4), np.float32)
    nn = np.zeros((max(n, 1), k), np.int32) if neighbours else None
    rc = L.orc_update_normals(capi.ptr(a, C.c_float), n, int(k), capi.ptr(vp, C.c_float), capi.ptr(out, C.c_float), capi.ptr(nn, C.c_int32))
    if rc != 0:
        raise RuntimeError(f"orc_update_normals rc={rc}")
    return (out[:n], nn[:n]) if neighbours else out[:n]


def make_keyframe_cloud(global_points, ids, min_grid_size, seed, pos0, orient0):
    L = lib()
    L.orc_make_keyframe_cloud.argtypes = [capi.c_float_p, capi.c_int32_p, C.c_int64, C.c_float, C.c_uint32, capi.c_double_p, capi.c_double_p, capi.c_float_p,
                                          capi.c_float_p, capi.c_int32_p, capi.c_int32_p, C.c_int64, capi.c_int64_p]
    a, ids = np.ascontiguousarray(global_points, np.float32), np.ascontiguousarray(ids, np.int32)
    n = a.shape[0]
    p, o = np.ascontiguousarray(pos0, np.float64), np.ascontiguousarray(orient0, np.float64)
    xyz, nrm = np.zeros((max(n, 1), 4), np.float32), np.zeros((max(n, 1), 4), np.float32)
    ring, src, m = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), C.c_int64(0)
    rc = L.orc_make_keyframe_cloud(capi.ptr(a, C.c_float), capi.ptr(ids, C.c_int32), n, float(np.float32(min_grid_size)), int(seed) & 0xFFFFFFFF,
                                   capi.ptr(p, C.c_double), capi.ptr(o, C.c_double), capi.ptr(xyz, C.c_float), capi.ptr(nrm, C.c_float), capi.ptr(ring, C.c_int32),
                                   capi.ptr(src, C.c_int32), n, C.byref(m))
    if rc != 0:
        raise RuntimeError(f"orc_make_keyframe_cloud rc={rc}")
    k = m.value
    return xyz[:k].copy(), nrm[:k].copy(), ring[:k].copy(), src[:k].copy()

