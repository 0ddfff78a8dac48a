"""Stage-by-stage comparison of a 'DMSAST03' dump of the REFERENCE (oracle/ref_harness/ref_main.cpp ... stage) with the oracle's
restatement.  Every check conditions on the reference's own result of the stage before it, so a failure names ONE statement of the
oracle -- and the hypothesis switch of oracle/dmsa_oracle.cpp (ORC_VAR_*) that would flip it:

  stage                 given (from the reference)      compared                         bar                      decides
  table                 start poses                     pose table                       <= 1 ulp, >= 90 % equal  GLIBC_TRIG; slerp / Floater-Hormann / Rodrigues restatements
  global_points         pose table                      transformed points               bit-exact                TRANSFORM_PAIRWISE; centralize's float subtraction
  member_lists          global points                   seg_offset, members, M1          bit-exact                PCL octree semantics, leaf acceptance, splitSet quirks
  fit_sums              global points + members         colwise().mean(), covariance     bit-exact                FIT_MEAN_TREE / FIT_FLOAT (mean); FIT_COV_TREE, the L1 size           
                                                        before limitCovariance,                                   behind Eigen's depth blocks (eigen_l1_bytes) for Gaussians above 680
                                                        pow(-1) of the counts                                     members (covariance); WEIGHT_DIV / the machine's libm (powf)
  eigen_solver          the reference's covariances     eigenvalues().real(),            bit-exact                LIMITCOV_JACOBI (any other eigen-decomposition), EIG_BACK_HALVES, EIG_NORMALIZE_SCALAR, the restatement of
                                                        eigenvectors().real()                                     Eigen 3.4.0's RealSchur / EigenSolver (oracle/eigensolver3f.h); libgcc's complex division
  info_mats             the reference's covariances,    information matrices, weights    bit-exact                LIMITCOV_VT (the rebuild V D V^-1), Matrix3f::inverse(); the weights' mean (VectorXf::mean)
                        members
  residuals             global points, info, weights    errorVec rows < M                bit-exact                SUM3_LEFT, MAHA_ASSOC, the float mean of :247-254
  jacobian              info, weights                   Jacobian                         5e-3 of the largest      the evaluation chain (pose chain, tables, transform) under a
                                                                                         entry (bit-exact with    forward difference; GLIBC_TRIG
                                                                                         the machine's libm)
  normal_equations      errorVec, Jacobian              H, raw step, clamp               1e-12 / 1e-9 relative    JTJ order (blocked sums, fma for P > 64), Gauss-Jordan vs PartialPivLU
  line_search           step, parameters                best_k, parameters after         exact                    adaptiveStepSize's arithmetic; strict '<'
Used by tests/test_ref_fixtures.py on the committed reference dumps (absent here: PARITY UNPINNED) and, as a check of the checks, on dumps the
oracle and its hypothesis variants write themselves."""
import os
import tempfile

import numpy as np

STAGES = ["table", "global_points", "member_lists", "fit_sums", "eigen_solver", "info_mats", "residuals", "jacobian", "normal_equations", "line_search"]
HINT = {
    "table": "GLIBC_TRIG (sin / cos / acos / atan2), or the slerp / Floater-Hormann / Rodrigues restatement",
    "global_points": "TRANSFORM_PAIRWISE (order of Matrix4f * Vector4f), or centralize()'s float subtraction",
    "member_lists": "PCL octree semantics (box growth, key generation, leaf order), leaf acceptance or the splitSet quirks -- no switch: a restatement bug",
    "fit_sums": "mean: FIT_MEAN_TREE / FIT_FLOAT; covariance: FIT_COV_TREE, or -- if only Gaussians above 680 members differ -- the L1 size behind "
                "Eigen's depth blocks (orc_set_eigen_l1_bytes / dmsa_debug_options::eigen_l1_bytes = the reference machine's L1d); pow(-1): WEIGHT_DIV, or another libm",
    "eigen_solver": "the restatement of EigenSolver<Matrix3f> (oracle/eigensolver3f.h: scaling, Hessenberg reflector, Francis QR steps, back substitution, back "
                    "transformation order EIG_BACK_HALVES, normalisation EIG_NORMALIZE_SCALAR) -- LIMITCOV_JACOBI is what ANY other eigen-decomposition looks like here; pairs: libgcc's __divsc3",
    "info_mats": "LIMITCOV_VT (V D V^T instead of V D V^-1) or the cofactor order of Matrix3f::inverse() once 'eigen_solver' has pinned the eigenpairs; weights: the order of "
                 "VectorXf::mean() (FIT_MEAN_TREE / FIT_FLOAT) once 'fit_sums' has pinned pow(-1)",
    "residuals": "SUM3_LEFT or MAHA_ASSOC (or the float mean of DmsaOptimizer.h:247-254)",
    "jacobian": "the evaluation chain under a forward difference: relative2global / pose tables (GLIBC_TRIG) / transform",
    "normal_equations": "JTJ_NOFMA / the blocked summation order of J^T J, or the Gauss-Jordan statement of H.inverse()",
    "line_search": "adaptiveStepSize's arithmetic (0.1 * k * step) or the strict '<' of the arg-min",
}


def _ulp_diff(a, b):
    """distance in float32 representable steps (same-sign finite values)"""
    ia = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    ib = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    return np.abs(ia - ib)


def _rows_of_points(prob, window):
    if window:
        return np.ascontiguousarray(prob.tformIdPerPoint, np.int32)
    return np.repeat(np.arange(prob.numFrames), np.diff(prob.frameOffsets)).astype(np.int32)


def _ids(prob, window):
    return np.concatenate([prob.ringIds, prob.staticRingIds]).astype(np.int32) if window else np.ascontiguousarray(prob.ringIds, np.int32)


class StageChecks:
    """ref: read_stage_dump() of the reference's file; prob / settings: the problem both sides started from."""

    def __init__(self, orc, ref, prob, settings, window):
        self.orc, self.ref, self.prob, self.s, self.window = orc, ref, prob, settings, window
        self.tmp = tempfile.mkdtemp(prefix="dmsa_stage_")
        self._own = None
        self.report = {}

    def own(self):
        if self._own is None:
            self._own = self.orc.stage_dump(self.prob, self.s, os.path.join(self.tmp, "own.bin"))
        return self._own

    def run(self, stage):
        getattr(self, "check_" + stage)()

    # ---- stages ----
    def check_table(self):
        a, b = self.own()["table"], self.ref["table"]
        assert a.shape == b.shape
        d = _ulp_diff(a, b)
        self.report["table"] = dict(equal=float(np.mean(d == 0)), max_ulp=int(d.max()))
        assert d.max() <= 1 and np.mean(d == 0) >= 0.9, self.report["table"]

    def check_global_points(self):
        ref, prob = self.ref, self.prob
        rows = _rows_of_points(prob, self.window)
        N = rows.shape[0]
        got = self.orc.transform_points(ref["table"], prob.localPoints, rows)
        assert np.array_equal(got[:, :3].view(np.int32), ref["global_xyz"][:N, :3].view(np.int32)), "transformed points differ GIVEN the reference's table"
        if self.window and ref["n"] > N:  # static tail: centralize() subtracts the first control translation in float (ContinuousTrajectory.h:75-88)
            assert np.array_equal(self.own()["global_xyz"][N:, :3].view(np.int32), ref["global_xyz"][N:, :3].view(np.int32)), "centralised static points differ"
        if not self.window:  # normals = R * n with the same table rows: compared where the table rows agree bit for bit
            same_row = np.all(self.own()["table"].view(np.int32) == ref["table"].view(np.int32), axis=1)[rows]
            assert np.array_equal(self.own()["global_normal"][same_row, :3].view(np.int32), ref["global_normal"][same_row, :3].view(np.int32)), "rotated normals differ"

    def _gaussians_on_ref_points(self):
        ref = self.ref
        return self.orc.Gaussians(ref["global_xyz"], _ids(self.prob, self.window), self.prob.minGridSize, self.s, normals4=ref["global_normal"])

    def check_member_lists(self):
        G, ref = self._gaussians_on_ref_points(), self.ref
        assert (G.M, G.M1, G.Mm) == (ref["M"], ref["M1"], ref["Mm"]), ((G.M, G.M1, G.Mm), (ref["M"], ref["M1"], ref["Mm"]))
        assert np.array_equal(G.seg_offset, ref["seg_offset"]) and np.array_equal(G.members, ref["members"])

    def check_fit_sums(self):
        """The fit's float reductions BEFORE limitCovariance / EigenSolver: mean, covariance and pow(-1) must be the reference's bits."""
        G, ref = self._gaussians_on_ref_points(), self.ref
        assert G.M == ref["M"], "member lists differ: fix that stage first"
        if ref.get("fit_mean") is None:
            raise AssertionError("the dump has no fit sums ('DMSAST01' of round 4): write it again with this round's ref_main.cpp")
        mean, cov, raw = G.fit_sums()
        counts = np.diff(G.seg_offset)
        eq_mean = np.all(mean.view(np.int32) == ref["fit_mean"].view(np.int32), axis=1)
        eq_cov = np.all(cov.view(np.int32) == ref["fit_cov"].view(np.int32), axis=1)
        eq_raw = raw.view(np.int32) == ref["weights_raw"].view(np.int32)
        big = counts > 680
        self.report["fit_sums"] = dict(means_bit_equal=float(np.mean(eq_mean)), covariances_bit_equal=float(np.mean(eq_cov)),
                                       covariances_bit_equal_up_to_680_members=float(np.mean(eq_cov[~big])) if np.any(~big) else 1.0,
                                       covariances_bit_equal_above_680_members=float(np.mean(eq_cov[big])) if np.any(big) else 1.0,
                                       pow_minus_one_bit_equal=float(np.mean(eq_raw)), counts_where_pow_differs=counts[~eq_raw][:8].tolist())
        assert eq_mean.all() and eq_cov.all() and eq_raw.all(), self.report["fit_sums"]

    @staticmethod
    def _same_bits(a, b):
        a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        return (a.view(np.int32) == b.view(np.int32)) | (np.isnan(a) & np.isnan(b))

    def check_eigen_solver(self):
        """EigenSolver<Matrix3f> on the REFERENCE's covariances (Gaussians.h:184-188): eigenvalues in the Schur form's order and the real parts of the
        normalised eigenvectors must be the reference's bits."""
        ref = self.ref
        if ref.get("eig_values") is None:
            raise AssertionError("the dump has no eigenpairs ('DMSAST02' of round 5): write it again with this round's ref_main.cpp")
        if ref.get("fit_cov") is None:
            raise AssertionError("the dump has no fit sums ('DMSAST01' of round 4): write it again with this round's ref_main.cpp")
        ev, V = self.orc.limitcov_eigenpairs(ref["fit_cov"])
        eq_val = np.all(self._same_bits(ev, ref["eig_values"]), axis=1)
        eq_vec = np.all(self._same_bits(V, ref["eig_vectors"]), axis=1)
        # the same eigenvalues in another order (what a symmetric solver would return) is reported apart from different values
        same_set = np.all(np.sort(ev, 1).view(np.int32) == np.sort(ref["eig_values"], 1).view(np.int32), axis=1)
        self.report["eigen_solver"] = dict(eigenvalues_bit_equal=float(np.mean(eq_val)), eigenvectors_bit_equal=float(np.mean(eq_vec)),
                                           eigenvalues_equal_as_a_set=float(np.mean(same_set)),
                                           max_rel_eigenvalue=float((np.abs(np.sort(ev, 1) - np.sort(ref["eig_values"], 1)).max(1) /
                                                                     np.maximum(np.abs(ref["eig_values"]).max(1), 1e-30)).max()))
        assert eq_val.all() and eq_vec.all(), self.report["eigen_solver"]

    def check_info_mats(self):
        """limitCovariance's rebuild and the inverse GIVEN the reference's covariances (the stage before pinned the eigenpairs), and the weights."""
        G, ref = self._gaussians_on_ref_points(), self.ref
        assert G.M == ref["M"], "member lists differ: fix that stage first"
        if ref.get("fit_cov") is None:
            raise AssertionError("the dump has no fit sums ('DMSAST01' of round 4): write it again with this round's ref_main.cpp")
        info = self.orc.info_from_covariance(ref["fit_cov"])
        eq = np.all(self._same_bits(info, ref["info"]), axis=1)
        scale = np.maximum(np.abs(ref["info"]).max(axis=1, keepdims=True), 1e-30)
        with np.errstate(invalid="ignore"):
            rel = np.nan_to_num(np.abs(info - ref["info"]) / scale, nan=0.0, posinf=0.0)
        wd = _ulp_diff(G.weights, ref["weights"])
        self.report["info_mats"] = dict(max_rel=float(rel.max()), matrices_bit_equal=float(np.mean(eq)),
                                        weights_max_ulp=int(wd.max()), weights_bit_equal=float(np.mean(wd == 0)))
        # every statement between the covariance and the information matrix is restated from Eigen's sources (EigenSolver, V D V^-1, cofactor inverse):
        # the reference's bits, no tolerance (round 6; rounds 1-5 allowed 1e-4 for a Jacobi stand-in); the weights likewise
        assert eq.all() and wd.max() == 0, self.report["info_mats"]

    def check_residuals(self):
        G, ref = self._gaussians_on_ref_points(), self.ref
        assert G.M == ref["M"], "member lists differ: fix that stage first"
        G.set_info(ref["info"], ref["weights"])
        e = G.residuals(ref["global_xyz"])
        M = ref["M"]
        bad = np.flatnonzero(e.view(np.int64) != ref["error_vec"][:M].view(np.int64))
        self.report["residuals"] = dict(rows_differing=int(bad.size), of=int(M))
        assert bad.size == 0, (self.report["residuals"], e[bad[:3]], ref["error_vec"][bad[:3]])
        if ref["a"] > 0:  # additional rows (gravity / odometry / IMU): double arithmetic on the poses
            add, radd = self.own()["error_vec"][-ref["a"]:], ref["error_vec"][M:]
            assert np.allclose(add, radd, rtol=1e-12, atol=1e-15), (add, radd)

    def check_jacobian(self):
        ref = self.ref
        own = self.orc.stage_dump(self.prob, self.s, os.path.join(self.tmp, "own_inj.bin"), inject_info=ref["info"], inject_weights=ref["weights"])
        assert own["jacobian"].shape == ref["jacobian"].shape, "member lists differ: fix that stage first"
        J, Jr = own["jacobian"], ref["jacobian"]
        scale = np.abs(Jr).max()
        self.report["jacobian"] = dict(max_rel_of_largest=float(np.abs(J - Jr).max() / scale), entries_bit_equal=float(np.mean(J.view(np.int64) == Jr.view(np.int64))))
        assert np.abs(J - Jr).max() <= 5e-3 * scale, self.report["jacobian"]

    def check_normal_equations(self):
        ref, s = self.ref, self.s
        H, g, step = self.orc.lm_step_from_jacobian(ref["error_vec"], ref["jacobian"], float(np.float32(s.lambda_diag)), s.step_length_optim)
        dH = np.abs(H - ref["H"]).max() / np.abs(ref["H"]).max()
        dS = np.abs(step - ref["step_raw"]).max() / np.abs(ref["step_raw"]).max()
        self.report["normal_equations"] = dict(H_rel=float(dH), step_rel=float(dS))
        assert dH <= 1e-12 and dS <= 1e-9, self.report["normal_equations"]
        raw = ref["step_raw"]
        max_elem = max(raw.max(), -raw.min())
        clamped = (s.max_step / max_elem) * raw if max_elem > s.max_step else raw
        assert np.array_equal(clamped, ref["step"]), "the clamp of DmsaOptimizer.h:125-128"

    def check_line_search(self):
        ref, own = self.ref, self.own()
        assert own["best_k"] == ref["best_k"], (own["best_k"], ref["best_k"])
        assert abs(own["error0"] - ref["error0"]) <= 1e-6 * ref["error0"], (own["error0"], ref["error0"])
        if ref["best_k"] > 0:  # params = raw + 0.1 * k * step (:160); raw = what getPoseParameters returned = own's start parameters
            raw = own["params_after"] - 0.1 * float(own["best_k"]) * own["step"]
            assert np.allclose(ref["params_after"], raw + 0.1 * float(ref["best_k"]) * ref["step"], rtol=0, atol=1e-12)
