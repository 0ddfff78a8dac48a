"""Symmetric 3 x 3 matrices for the EigenSolver<Matrix3f> tests (Gaussians.h:184-188): sample covariances of every shape a voxel can have,
incl. the degenerate ones (planar, collinear, repeated eigenvalues, isotropic, tiny, exact zeros)."""
import numpy as np


def covariance_cases(seed: int, per_kind: int):
    rng = np.random.default_rng(seed)
    out = {}

    def build(lam):
        Q, _ = np.linalg.qr(rng.normal(size=(lam.shape[0], 3, 3)))
        A = np.einsum("nij,nj,nkj->nik", Q, lam, Q).astype(np.float32)
        return ((A + A.transpose(0, 2, 1)) * np.float32(0.5)).astype(np.float32)  # exactly symmetric, as centered^T * centered is

    n = per_kind
    out["generic"] = build(rng.uniform(1e-4, 2.0, (n, 3)))
    out["planar"] = build(np.stack([rng.uniform(0.01, 1, n), rng.uniform(0.01, 1, n), rng.uniform(1e-9, 1e-5, n)], 1))
    out["line"] = build(np.stack([rng.uniform(0.01, 1, n), rng.uniform(1e-10, 1e-6, n), rng.uniform(1e-10, 1e-6, n)], 1))
    a = rng.uniform(0.01, 1, n)
    out["repeated"] = build(np.stack([a, a, rng.uniform(0.001, 1, n)], 1))
    out["isotropic"] = build(np.stack([a, a, a], 1))
    out["tiny"] = build(rng.uniform(1e-12, 1e-8, (n, 3)))
    out["large"] = build(rng.uniform(1e3, 1e8, (n, 3)))
    # sample covariances of actual small point sets (what the fit feeds the solver), float arithmetic
    pts = rng.normal(size=(n, 12, 3)).astype(np.float32) * rng.uniform(0.001, 0.5, (n, 1, 3)).astype(np.float32) + rng.uniform(-50, 50, (n, 1, 3)).astype(np.float32)
    c = pts - pts.mean(1, keepdims=True, dtype=np.float32)
    out["samples"] = (np.einsum("nki,nkj->nij", c, c) / np.float32(11)).astype(np.float32)
    # structured exact cases: diagonal, one off-diagonal pair, zeros, a 2 x 2 block, already Hessenberg, axis-aligned planes
    ex = []
    for d in ([1, 2, 3], [3, 2, 1], [1, 1, 1], [0, 0, 0], [1e-6, 1, 1e-6], [5, 0, 0], [0, 0, 7]):
        ex.append(np.diag(np.array(d, np.float32)))
    for (i, j) in ((0, 1), (0, 2), (1, 2)):
        m = np.diag(np.array([2, 3, 4], np.float32))
        m[i, j] = m[j, i] = 0.5
        ex.append(m)
        m = np.zeros((3, 3), np.float32)
        m[i, j] = m[j, i] = 1.0
        ex.append(m)
    ex.append(np.ones((3, 3), np.float32))
    ex.append(np.full((3, 3), 1e-30, np.float32))
    ex.append(np.full((3, 3), 1e-42, np.float32))  # denormal: below the solver's "considerAsZero"
    out["exact"] = np.stack(ex).astype(np.float32)
    return out
