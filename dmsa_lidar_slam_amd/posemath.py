"""Small SE(3) helpers used by the synthetic-data generator and the keyframe sharding glue
(not on the per-iteration hot path).  Semantics follow ConsecutivePoses.h:26-67:
global pose k = chain of relative poses 0..k with T_k = T_{k-1} + R_{k-1} t_k, R_k = R_{k-1} exp(w_k)."""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation as Rot


def relative2global(rel_o: np.ndarray, rel_t: np.ndarray):
    n = rel_o.shape[0]
    go = np.zeros((n, 3))
    gt = np.zeros((n, 3))
    R = np.eye(3)
    T = np.zeros(3)
    for k in range(n):
        T = T + R @ rel_t[k]
        gt[k] = T
        R = R @ Rot.from_rotvec(rel_o[k]).as_matrix()
        go[k] = Rot.from_matrix(R).as_rotvec()
    return go, gt


def global2relative(go: np.ndarray, gt: np.ndarray):
    n = go.shape[0]
    ro = np.zeros((n, 3))
    rt = np.zeros((n, 3))
    ro[0], rt[0] = go[0], gt[0]
    for k in range(n - 1, 0, -1):
        R1 = Rot.from_rotvec(go[k - 1]).as_matrix()
        R2 = Rot.from_rotvec(go[k]).as_matrix()
        ro[k] = Rot.from_matrix(R1.T @ R2).as_rotvec()
        rt[k] = R1.T @ (gt[k] - gt[k - 1])
    return ro, rt
