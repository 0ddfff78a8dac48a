"""Python mirror of keyframe creation over include/dmsa_keyframe_cloud.h: DmsaSlam::updateNormals
(DmsaSlam.h:553-567, pcl::NormalEstimationOMP with k = 6) and the cloud part of addNewKeyframeToMap (:497-531), on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from .api import DmsaError
from .static_points import _xyz4


class KeyframeCloudBuilder:
    def __init__(self, device: int = 0, optimizer=None):
        """optimizer: a DmsaOptimizer whose context (and resident window cloud) this builder shares; None = own context."""
        self._lib = capi.load_library()
        self._owner = optimizer is None
        if optimizer is not None:
            self._ctx = optimizer._ctx
            return
        ctx = C.c_void_p()
        rc = self._lib.dmsa_create(device, 0, C.byref(ctx))
        if rc != capi.DMSA_OK:
            raise DmsaError(f"dmsa_create failed with {rc}: keyframe creation runs on the GPU, there is no CPU fallback")
        self._ctx = ctx

    def close(self):
        if getattr(self, "_ctx", None) and self._owner:
            self._lib.dmsa_destroy(self._ctx)
        self._ctx = None

    __del__ = close

    def _check(self, rc, what):
        if rc != capi.DMSA_OK:
            raise DmsaError(f"{what} failed with {rc}: {self._lib.dmsa_last_error(self._ctx).decode()}")

    def updateNormals(self, cloud, cellHint: float, k: int = 6, origin=(0.0, 0.0, 0.0), neighbours: bool = False):
        """Returns normals (n,4) = (normal_x, normal_y, normal_z, curvature) [, neighbour indices (n,k) in search order]."""
        a = _xyz4(cloud)
        n = a.shape[0]
        vp = np.ascontiguousarray(origin, np.float32)
        out = np.zeros((max(n, 1), 4), np.float32)
        nn = np.zeros((max(n, 1), k), np.int32) if neighbours else None
        self._check(self._lib.dmsa_update_normals(self._ctx, capi.ptr(a, C.c_float), n, int(k), float(np.float32(cellHint)), capi.ptr(vp, C.c_float),
                                                  capi.ptr(out, C.c_float), capi.ptr(nn, C.c_int32)), "dmsa_update_normals")
        return (out[:n], nn[:n]) if neighbours else out[:n]

    def addNewKeyframeCloud(self, globalPoints, ids, minGridSize: float, seed: int, pos0, orient0, numResident: int = 0):
        """addNewKeyframeToMap (:497-531) without the bookkeeping: (pointCloudLocal xyz (m,4), normals (m,4), ringIds (m,), index into
        globalPoints (m,)).  globalPoints None: the first numResident global points / ring ids resident in the shared optimizer context."""
        a = _xyz4(globalPoints) if globalPoints is not None else None
        ids = np.ascontiguousarray(ids, np.int32) if globalPoints is not None else None
        n = a.shape[0] if a is not None else int(numResident)
        p, o = np.ascontiguousarray(pos0, np.float64), np.ascontiguousarray(orient0, np.float64)
        xyz, nrm = np.zeros((max(n, 1), 4), np.float32), np.zeros((max(n, 1), 4), np.float32)
        ring, src, m = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32), C.c_int64(0)
        self._check(self._lib.dmsa_make_keyframe_cloud(self._ctx, capi.ptr(a, C.c_float), capi.ptr(ids, C.c_int32), n, float(np.float32(minGridSize)),
                                                       int(seed) & 0xFFFFFFFF, capi.ptr(p, C.c_double), capi.ptr(o, C.c_double), capi.ptr(xyz, C.c_float),
                                                       capi.ptr(nrm, C.c_float), capi.ptr(ring, C.c_int32), capi.ptr(src, C.c_int32), n, C.byref(m)),
                    "dmsa_make_keyframe_cloud")
        k = m.value
        return xyz[:k].copy(), nrm[:k].copy(), ring[:k].copy(), src[:k].copy()
