"""Parity against the REAL reference, statement by statement.  scripts/build_ref_oracle.sh (needs Eigen 3.4 / PCL 1.10 / Boost / ROS headers -- absent from the
graft image) runs davidskdds/DMSA_LiDAR_SLAM's own DmsaOptimizer::optimizeSet on the seeded problems of tests/golden/make_ref_inputs.py
and writes tests/golden/ref_<case>.poses.bin.  While those files are absent the oracle stays "parity unpinned" and these tests SKIP,
loudly; once committed, the CPU oracle (and the HIP library on a GPU) must reproduce the reference's poses within BASELINE.json's
1e-4 m / 1e-4 rad."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

import make_ref_inputs as ref_inputs  # noqa: E402
import ref_stage_checks as stages  # noqa: E402
from dmsa_lidar_slam_amd import dump  # noqa: E402
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings  # noqa: E402

TOL = 1e-4


def _ref_l1():
    """L1d size of the machine the reference fixtures were produced on (scripts/build_ref_oracle.sh writes tests/golden/ref_machine.json): Eigen
    sizes the depth blocks of centered^T * centered from it, so the oracle / the library are told the same number."""
    import json
    path = os.path.join(HERE, "golden", "ref_machine.json")
    if not os.path.exists(path):
        return 32 * 1024
    return int(json.load(open(path)).get("eigen_l1_bytes", 32 * 1024))


@pytest.fixture
def ref_orc(orc):
    orc.set_eigen_l1_bytes(_ref_l1())
    yield orc
    orc.set_eigen_l1_bytes(32 * 1024)


def _fixture(name):
    path = os.path.join(HERE, "golden", f"ref_{name}.poses.bin")
    if not os.path.exists(path):
        pytest.skip(f"PARITY UNPINNED: {os.path.relpath(path)} absent -- run scripts/build_ref_oracle.sh on a machine that can build the reference")
    return dump.read_poses(path)


def _settings(name):
    it = ref_inputs.ITERATIONS[name]
    return DmsaOptimSettings.keyframe_map(num_iter=it) if name.startswith("keyframes") else DmsaOptimSettings.sliding_window(num_iter=it)


def _global(orc, ro, rt):
    return orc.relative2global(ro, rt)


@pytest.mark.parametrize("name", list(ref_inputs.CASES))
def test_oracle_reproduces_the_reference(ref_orc, name):
    orc = ref_orc
    ro_ref, rt_ref = _fixture(name)
    prob = ref_inputs.CASES[name]()
    (orc.optimize_keyframes if name.startswith("keyframes") else orc.optimize_window)(prob, _settings(name))
    go, gt = _global(orc, prob.relOrientations, prob.relTranslations)
    go_r, gt_r = _global(orc, ro_ref, rt_ref)
    assert np.abs(gt - gt_r).max() < TOL and np.abs(go - go_r).max() < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(ref_inputs.CASES))
def test_hip_library_reproduces_the_reference(hip, orc, name):
    ro_ref, rt_ref = _fixture(name)
    prob = ref_inputs.CASES[name]()
    hip.DmsaOptimizer(debug={"eigen_l1_bytes": _ref_l1()}).optimizeSet(prob, _settings(name))
    go, gt = _global(orc, prob.relOrientations, prob.relTranslations)
    go_r, gt_r = _global(orc, ro_ref, rt_ref)
    assert np.abs(gt - gt_r).max() < TOL and np.abs(go - go_r).max() < TOL


# ---- iteration 0, stage by stage (tests/ref_stage_checks.py names what every stage decides) -------------------------------------------
def _stage_fixture(name):
    path = os.path.join(HERE, "golden", f"ref_{name}.stage.bin")
    if not os.path.exists(path):
        pytest.skip(f"PARITY UNPINNED: {os.path.relpath(path)} absent -- run scripts/build_ref_oracle.sh on a machine that can build the reference")
    return dump.read_stage_dump(path)


def _one_iteration(name):
    return DmsaOptimSettings.keyframe_map(num_iter=1) if name.startswith("keyframes") else DmsaOptimSettings.sliding_window(num_iter=1)


@pytest.mark.parametrize("stage", stages.STAGES)
@pytest.mark.parametrize("name", list(ref_inputs.CASES))
def test_oracle_stage_matches_the_reference(ref_orc, name, stage):
    orc = ref_orc
    ref = _stage_fixture(name)
    chk = stages.StageChecks(orc, ref, ref_inputs.CASES[name](), _one_iteration(name), not name.startswith("keyframes"))
    try:
        chk.run(stage)
    except AssertionError as e:
        raise AssertionError(f"stage '{stage}' of {name} differs from the reference -- decides: {stages.HINT[stage]}\n{e}") from e
    if stage in chk.report:
        print(f"[ref stage] {name} {stage}: {chk.report[stage]}")


@pytest.mark.parametrize("name", list(ref_inputs.CASES))
def test_oracle_poses_after_one_iteration(ref_orc, name):
    orc = ref_orc
    path = os.path.join(HERE, "golden", f"ref_{name}.iter1.poses.bin")
    if not os.path.exists(path):
        pytest.skip(f"PARITY UNPINNED: {os.path.relpath(path)} absent")
    ro_ref, rt_ref = dump.read_poses(path)
    prob = ref_inputs.CASES[name]()
    (orc.optimize_keyframes if name.startswith("keyframes") else orc.optimize_window)(prob, _one_iteration(name))
    go, gt = _global(orc, prob.relOrientations, prob.relTranslations)
    go_r, gt_r = _global(orc, ro_ref, rt_ref)
    assert np.abs(gt - gt_r).max() < 1e-5 and np.abs(go - go_r).max() < 1e-5, (np.abs(gt - gt_r).max(), np.abs(go - go_r).max())


# ---- a check of the checks: dumps the oracle and its hypothesis variants write themselves ------------------------------------------------
def _variant_dump(variant, name, path):
    """the stage dump of `name` written by the oracle built with -DORC_VAR_<variant> (own process: the library is loaded once per process)"""
    import subprocess

    root = os.path.dirname(HERE)
    lib = os.path.join(root, "oracle", "_variants", f"libdmsa_oracle_{variant}.so")
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "-s", f"_variants/libdmsa_oracle_{variant}.so"])
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import make_ref_inputs as r; from oracle import oracle_py as o; "
            "from dmsa_lidar_slam_amd.problems import DmsaOptimSettings as S; "
            "s = S.keyframe_map(num_iter=1) if %r.startswith('keyframes') else S.sliding_window(num_iter=1); o.stage_dump(r.CASES[%r](), s, %r)"
            % (root, os.path.join(HERE, "golden"), name, name, path))
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, DMSA_ORACLE_LIB=lib))
    return dump.read_stage_dump(path)


@pytest.mark.parametrize("name", list(ref_inputs.CASES))
def test_stage_checks_pass_on_the_oracles_own_dump(orc, name, tmp_path):
    """NOT a pin (the oracle against itself): it shows that the stage checks, the dump layout and the conditional re-evaluations are
    consistent, so that a failure on a real reference dump means a difference and not a bug of the test."""
    s = _one_iteration(name)
    ref = orc.stage_dump(ref_inputs.CASES[name](), s, str(tmp_path / "self.bin"))
    chk = stages.StageChecks(orc, ref, ref_inputs.CASES[name](), s, not name.startswith("keyframes"))
    for stage in stages.STAGES:
        chk.run(stage)
    assert chk.report["table"]["equal"] == 1.0 and chk.report["info_mats"]["matrices_bit_equal"] == 1.0 and chk.report["jacobian"]["entries_bit_equal"] == 1.0


@pytest.mark.parametrize("variant,fails,passes", [
    ("TRANSFORM_PAIRWISE", ["global_points"], ["table", "member_lists", "fit_sums", "eigen_solver", "info_mats", "residuals", "normal_equations"]),
    ("SUM3_LEFT", ["residuals"], ["table", "global_points", "member_lists", "fit_sums", "eigen_solver", "normal_equations"]),
    ("MAHA_ASSOC", ["residuals"], ["table", "global_points", "member_lists", "fit_sums", "eigen_solver", "info_mats", "normal_equations"]),
    ("FIT_FLOAT", ["fit_sums", "info_mats"], ["table", "global_points", "member_lists", "eigen_solver", "residuals", "normal_equations"]),
    ("FIT_MEAN_TREE", ["fit_sums"], ["table", "global_points", "member_lists", "eigen_solver", "residuals", "normal_equations"]),
    ("FIT_COV_TREE", ["fit_sums"], ["table", "global_points", "member_lists", "eigen_solver", "info_mats", "residuals", "normal_equations"]),
    ("LIMITCOV_VT", ["info_mats"], ["table", "global_points", "member_lists", "fit_sums", "eigen_solver", "residuals", "normal_equations"]),
    ("LIMITCOV_JACOBI", ["eigen_solver", "info_mats"], ["table", "global_points", "member_lists", "fit_sums", "residuals", "normal_equations"]),
    ("EIG_BACK_HALVES", ["eigen_solver"], ["table", "global_points", "member_lists", "fit_sums", "residuals", "normal_equations"]),
    ("EIG_NORMALIZE_SCALAR", ["eigen_solver"], ["table", "global_points", "member_lists", "fit_sums", "residuals", "normal_equations"]),
])
def test_every_float_order_hypothesis_is_decided_by_one_stage(orc, tmp_path, variant, fails, passes):
    """A dump written by the oracle built with ONE alternative reading stands in for "the reference turned out to evaluate it the other
    way": exactly the stage that owns the statement fails (stages condition on the reference's result of the stage before), the stages
    before and the independent ones pass.  The fit's summation orders are decided bit for bit by 'fit_sums' (mean, covariance before
    limitCovariance, pow(-1) of the counts -- computed by the harness with the reference's own Eigen expressions); FIT_FLOAT also fails
    'info_mats' through the weights (their mean as a scalar chain is ulps away from Eigen's redux order).  Round 6: the eigen-decomposition has its
    own stage ('eigen_solver', on the reference's covariances: LIMITCOV_JACOBI -- any solver that is not Eigen's -- and the order of its one 3-term
    sum fail there), and 'info_mats' is bit-exact given those covariances, so LIMITCOV_VT now FAILS it instead of hiding inside a 1e-4 bound."""
    name = "window_static"
    ref = _variant_dump(variant, name, str(tmp_path / f"{variant}.bin"))
    s = _one_iteration(name)
    chk = stages.StageChecks(orc, ref, ref_inputs.CASES[name](), s, True)
    for stage in passes:
        chk.run(stage)
    for stage in fails:
        with pytest.raises(AssertionError):
            chk.run(stage)
    if variant == "FIT_FLOAT":
        assert chk.report["info_mats"]["weights_max_ulp"] > 0 and chk.report["residuals"]["rows_differing"] == 0
    if variant == "FIT_MEAN_TREE":
        assert chk.report["fit_sums"]["means_bit_equal"] < 1.0 and chk.report["fit_sums"]["pow_minus_one_bit_equal"] == 1.0
    if variant == "FIT_COV_TREE":
        assert chk.report["fit_sums"]["means_bit_equal"] == 1.0 and chk.report["fit_sums"]["covariances_bit_equal"] < 1.0, chk.report["fit_sums"]
    if variant == "LIMITCOV_VT":
        assert chk.report["info_mats"]["matrices_bit_equal"] < 0.5 and chk.report["info_mats"]["weights_bit_equal"] == 1.0, chk.report["info_mats"]
        assert chk.report["info_mats"]["max_rel"] < 1e-4  # ... which the bound of rounds 1-5 would have let through
    if variant == "LIMITCOV_JACOBI":
        r = chk.report["eigen_solver"]
        assert r["eigenvalues_bit_equal"] < 0.5 and r["max_rel_eigenvalue"] < 1e-4, r  # the same mathematics, other order / other last bits
    if variant == "EIG_BACK_HALVES":
        r = chk.report["eigen_solver"]
        assert r["eigenvalues_bit_equal"] == 1.0 and r["eigenvectors_bit_equal"] < 1.0, r  # only the last eigenvector's back transformation moves
    if variant == "EIG_NORMALIZE_SCALAR":
        r = chk.report["eigen_solver"]
        assert r["eigenvalues_bit_equal"] == 1.0 and r["eigenvectors_bit_equal"] < 1.0, r  # rows 0 and 1 of the normalised eigenvectors move by an ulp


def test_a_gaussian_count_where_powf_is_not_the_division_is_decided_by_fit_sums(orc):
    """libm's powf(n, -1.0f) is not correctly rounded: for some counts (953, 2071, 5331, ... with glibc >= 2.27) it is one ulp away from
    1.0f / n.  The oracle's pow(-1) is the libm call (Gaussians.h:172 through Eigen's scalar_pow_op); the alternative reading WEIGHT_DIV is
    the division.  A Gaussian with such a count separates the two in the 'fit_sums' stage."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.powf.argtypes, libm.powf.restype = [C.c_float, C.c_float], C.c_float
    n = np.arange(1, 20000, dtype=np.float32)
    pw = np.array([libm.powf(float(v), -1.0) for v in n], np.float32)
    div = (np.float32(1.0) / n).astype(np.float32)
    differing = n[pw.view(np.int32) != div.view(np.int32)].astype(int)
    if differing.size == 0:
        pytest.skip("this machine's powf(n, -1) equals 1 / n for every n < 20000")
    cnt = int(differing[0])
    # one leaf with exactly `cnt` points, plus a second one
    rng = np.random.default_rng(5)
    # (PCL's lattice is anchored at the first point: cell faces at p0 + k * resolution, so p0 = 0 puts both clusters inside one cell of either level)
    pts = np.zeros((cnt + 40, 4), np.float32)
    pts[1:40, :3] = rng.uniform(0.05, 0.25, (39, 3))
    pts[40:, :3] = rng.uniform(5.05, 5.25, (cnt, 3))
    ids = (np.arange(cnt + 40) % 7).astype(np.int32)
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    s = DmsaOptimSettings.sliding_window()
    G = orc.Gaussians(pts, ids, 0.5, s)
    _, _, raw = G.fit_sums()
    counts = np.diff(G.seg_offset)
    assert cnt in counts.tolist(), counts
    k = int(np.flatnonzero(counts == cnt)[0])
    assert raw[k].view(np.int32) == pw[cnt - 1].view(np.int32) and raw[k].view(np.int32) != div[cnt - 1].view(np.int32)


def test_ref_inputs_are_deterministic(tmp_path):
    """the dumps the reference harness reads are a pure function of the committed seeds"""
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    dump.write_window_problem(a, ref_inputs.CASES["window_small"]())
    dump.write_window_problem(b, ref_inputs.CASES["window_small"]())
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a, "rb").read(8) == b"DMSAWN01"
