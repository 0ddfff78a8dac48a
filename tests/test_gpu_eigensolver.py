"""GPU parity of Gaussians::limitCovariance / EigenSolver<Matrix3f> (Gaussians.h:181-201): the device's hand-specialised 3 x 3 solver
(csrc/dmsa_kernels.hip: eigensolver3f) against the oracle's statement-by-statement restatement of Eigen 3.4.0 (oracle/eigensolver3f.h) --
two independent writings of the same algorithm -- bit for bit on 10^5 matrices of every shape, incl. the complex-pair corner."""
import numpy as np
import pytest

from eig_cases import covariance_cases

pytestmark = pytest.mark.gpu


def test_device_solver_has_the_oracles_bits(hip, orc):
    opt = hip.DmsaOptimizer(device=0)
    total = 0
    for kind, A in covariance_cases(5, 16000).items():
        re, im, V, it, info, pairs = orc.eigensolver3f(A)
        L = orc.limit_covariance(A)
        Ld, ev, Vd, itd, infod = opt.limitCovariance(A)
        assert np.array_equal(infod, info) and np.array_equal(itd, it), kind
        assert np.array_equal(ev.view(np.int32), re.view(np.int32)), kind
        assert np.array_equal(Vd.view(np.int32), V.view(np.int32)), (kind, int((Vd.view(np.int32) != V.view(np.int32)).any(axis=(1, 2)).sum()))
        same = (Ld.view(np.int32) == L.view(np.int32)) | (np.isnan(Ld) & np.isnan(L))
        assert same.all(), (kind, int((~same).any(axis=(1, 2)).sum()), int(pairs.sum()))
        total += A.shape[0]
    assert total >= 100000


def test_non_finite_input_takes_the_same_way_out(hip, orc):
    A = covariance_cases(6, 8)["generic"].copy()
    A[0, 0, 0] = np.nan
    A[1, :, :] = np.nan
    A[2, 1, 2] = A[2, 2, 1] = np.inf
    L = orc.limit_covariance(A)
    Ld, _, _, _, infod = hip.DmsaOptimizer(device=0).limitCovariance(A)
    _, _, _, _, info, _ = orc.eigensolver3f(A)
    assert np.array_equal(infod, info)
    assert np.array_equal(np.isnan(Ld), np.isnan(L)) and np.array_equal(Ld[np.isfinite(L)].view(np.int32), L[np.isfinite(L)].view(np.int32))
