"""EigenSolver<Matrix3f> (Gaussians.h:184-188) as the oracle restates Eigen 3.4.0's algorithm (oracle/eigensolver3f.h: scaling, Householder
reduction to Hessenberg form, Francis double-shift QR to the real Schur form, back substitution, back transformation, normalised columns).
Eigen is not in this image, so the BITS stay unpinned until scripts/build_ref_oracle.sh runs somewhere; what IS checked here is the
mathematics against numpy on 10^5 matrices of every shape, that the iteration caps (40 per row, the exceptional shifts at 10 / 30) are
never reached, and the solver's documented corner: a complex-conjugate pair out of a block of rounding noise makes eigenvectors().real()
singular -- the reference's limitCovariance then produces a non-finite covariance, and so does the restatement."""
import os
import subprocess

import numpy as np

from eig_cases import covariance_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eigenpairs_against_numpy(orc):
    total = pairs_total = 0
    for kind, A in covariance_cases(0, 14000).items():
        re, im, V, it, info, pairs = orc.eigensolver3f(A)
        assert (info == 0).all(), kind
        assert it.max() <= 9, (kind, int(it.max()))  # the exceptional shift of iteration 10 is never reached, let alone the cap of 120
        scale = np.maximum(np.abs(A).max(axis=(1, 2)), np.float32(1e-38)).astype(np.float64)
        nz = np.abs(A).max(axis=(1, 2)) >= np.finfo(np.float32).tiny  # below FLT_MIN the solver returns T = 0, U = I by design (test_exact_cases)
        real = (pairs == 0) & nz
        w = np.linalg.eigvalsh(A.astype(np.float64))
        err = np.abs(np.sort(re.astype(np.float64), 1) - w).max(1) / scale
        assert err[real].max() < 4e-6, (kind, float(err[real].max()))
        Ad, Vd = A.astype(np.float64), V.astype(np.float64)
        resid = np.abs(np.einsum("nij,njk->nik", Ad, Vd) - Vd * re[:, None, :].astype(np.float64)).max(axis=(1, 2)) / scale
        assert resid[real].max() < 4e-6, (kind, float(resid[real].max()))
        # columns are unit vectors, the imaginary parts are zero wherever no pair was left
        nrm = np.sqrt((Vd ** 2).sum(1))
        assert np.abs(nrm[real] - 1).max() < 1e-6 and not im[real].any()
        total += A.shape[0]
        pairs_total += int(pairs.sum())
        if kind in ("generic", "planar", "samples", "large", "tiny", "isotropic"):
            assert pairs.sum() == 0, kind
    assert total >= 100000
    assert 0 < pairs_total < 200  # only the collinear / repeated shapes leave a 2 x 2 block of noise now and then


def test_limit_covariance_against_eigh(orc):
    for kind, A in covariance_cases(1, 4000).items():
        if kind == "exact":
            continue
        _, _, _, _, _, pairs = orc.eigensolver3f(A)
        L = orc.limit_covariance(A)
        w, v = np.linalg.eigh(A.astype(np.float64))
        ref = np.einsum("nij,nj,nkj->nik", v, np.maximum(w, 1e-4), v)
        scale = np.maximum(np.abs(A).max(axis=(1, 2)), 1e-4)
        d = np.abs(L - ref).max(axis=(1, 2)) / scale
        ok = pairs == 0
        assert np.isfinite(L[ok]).all() and d[ok].max() < 2e-5, (kind, float(d[ok].max()))
        # the documented corner: a pair -> two equal real parts -> singular V -> not finite (Gaussians.h:188, :200)
        assert not np.isfinite(L[~ok]).all(axis=(1, 2)).any()


def test_exact_cases(orc):
    A = covariance_cases(2, 4)["exact"]
    re, im, V, it, info, pairs = orc.eigensolver3f(A)
    assert (info == 0).all() and (pairs == 0).all()
    # a diagonal matrix is its own Schur form: eigenvalues in place, eigenvectors the unit vectors, no iteration
    for k, d in enumerate(([1, 2, 3], [3, 2, 1], [1, 1, 1], [0, 0, 0], [1e-6, 1, 1e-6], [5, 0, 0], [0, 0, 7])):
        assert np.array_equal(re[k], np.array(d, np.float32)) and it[k] == 0
        assert np.array_equal(np.abs(V[k]), np.eye(3, dtype=np.float32))
    # below FLT_MIN the solver returns T = 0, U = I; limitCovariance then rebuilds 1e-4 * I
    L = orc.limit_covariance(A[-1:])
    assert np.array_equal(L[0], np.float32(1e-4) * np.eye(3, dtype=np.float32))
    assert np.array_equal(re[-1], np.zeros(3, np.float32))


def test_jacobi_statement_of_rounds_1_to_5_is_the_same_mathematics():
    """ORC_VAR_LIMITCOV_JACOBI (the oracle's statement until round 6) differs from the restated EigenSolver in eigenvalue order and in the last bits only."""
    import ctypes as C

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_variants/libdmsa_oracle_LIMITCOV_JACOBI.so"])
    lj = C.CDLL(os.path.join(ROOT, "oracle", "_variants", "libdmsa_oracle_LIMITCOV_JACOBI.so"))
    from oracle import oracle_py as orc

    fp = C.POINTER(C.c_float)
    lj.orc_limit_covariance.argtypes, lj.orc_limit_covariance.restype = [fp, C.c_int64, fp], None
    worst = 0.0
    differ = 0
    for kind in ("generic", "planar", "samples"):
        A = covariance_cases(3, 3000)[kind]
        At = np.ascontiguousarray(A.transpose(0, 2, 1))
        out = np.zeros_like(At)
        lj.orc_limit_covariance(At.ctypes.data_as(fp), A.shape[0], out.ctypes.data_as(fp))
        J = out.transpose(0, 2, 1)
        E = orc.limit_covariance(A)
        scale = np.maximum(np.abs(A).max(axis=(1, 2)), 1e-4)
        worst = max(worst, float((np.abs(J - E).max(axis=(1, 2)) / scale).max()))
        differ += int((J != E).any(axis=(1, 2)).sum())
    assert worst < 2e-5 and differ > 1000  # same covariance to float accuracy, almost never the same bits
