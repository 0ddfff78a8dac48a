"""A tolerance-based witness of the fit and of the residuals that does NOT share the oracle's summation orders (ADVICE, round 5): the golden
fixtures and the GPU parity tests prove device == oracle, and the oracle's float orders are restated from Eigen's sources; if one of those
recollections were wrong in a way that changes the MATHEMATICS (not the last bits), every bit-level test would still pass.  So: mean,
covariance, eigenvalue clamp, inverse and the Mahalanobis residuals once more in plain numpy double precision, order-independent, against
the oracle's float results with bounds that follow from float accuracy (condition of the clamped covariance), on a window and on a keyframe set."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings


def _cases():
    w = synth.window_problem(seed=11, scans=3, rings=32, az_steps=256, num_static=4000)
    k = synth.keyframe_problem(seed=3, frames=6, rings=16, az_steps=128, arc=0.5)
    return [("window", w, DmsaOptimSettings.sliding_window()), ("keyframes", k, DmsaOptimSettings.keyframe_map())]


@pytest.mark.parametrize("case", [0, 1])
def test_fit_and_residuals_against_double_precision_numpy(orc, case):
    name, prob, s = _cases()[case]
    window = name == "window"
    dumped = orc.stage_dump(prob.copy(), s, f"/tmp/_fit_sanity_{name}.bin")
    xyz = dumped["global_xyz"]
    ids = np.concatenate([prob.ringIds, prob.staticRingIds]).astype(np.int32) if window else np.ascontiguousarray(prob.ringIds, np.int32)
    G = orc.Gaussians(xyz, ids, prob.minGridSize, s, normals4=dumped["global_normal"])
    assert G.M == dumped["M"] and G.M > 100
    e = G.residuals(xyz)
    pts = xyz[:, :3].astype(np.float64)
    worst_info = worst_e = 0.0
    skipped = 0
    for g in range(G.M):
        m = G.members[G.seg_offset[g]:G.seg_offset[g + 1]]
        x = pts[m]
        n = x.shape[0]
        mean = x.mean(0)
        c = x - mean
        cov = c.T @ c / (n - 1)
        w, v = np.linalg.eigh(cov)
        if w[0] < 1e-4 * 1.001 and w[0] > 1e-4 * 0.999:
            skipped += 1  # an eigenvalue within float accuracy of the clamp: either side is right
            continue
        cl = (v * np.maximum(w, 1e-4)) @ v.T
        info = np.linalg.inv(cl)
        got = G.info[g].reshape(3, 3).T.astype(np.float64)  # column-major
        cond = max(w.max(), 1e-4) / 1e-4
        rel = np.abs(got - info).max() / np.abs(info).max()
        # float eigen-decomposition + inverse of a matrix of condition `cond`: a few hundred eps * cond; centred float coordinates add eps * |x|^2 / lambda
        bound = 400 * np.finfo(np.float32).eps * cond + 40 * np.finfo(np.float32).eps * (np.abs(x).max() ** 2) / 1e-4 * 1e-2
        assert rel < max(bound, 1e-4), (g, n, rel, bound, w)
        worst_info = max(worst_info, rel)
        # the residual of DmsaOptimizer.h:247-267 with the oracle's OWN information matrix (so that only the sums are compared)
        d = x - x.astype(np.float32).mean(0, dtype=np.float64)
        q = float(G.weights[g]) * np.einsum("ij,jk,ik->", d, got, d)
        ref_e = np.sqrt(abs(q))
        re = abs(e[g] - ref_e) / max(ref_e, 1e-12)
        assert re < 1e-5, (g, n, e[g], ref_e)
        worst_e = max(worst_e, re)
    assert skipped < 0.05 * G.M
    print(f"[fit sanity] {name}: M = {G.M}, worst relative information-matrix error {worst_info:.2e}, worst relative residual error {worst_e:.2e}")
