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
def test_oracle_reproduces_the_reference(orc, name):
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
    hip.DmsaOptimizer().optimizeSet(prob, _settings(name))
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
def test_oracle_stage_matches_the_reference(orc, name, stage):
    ref = _stage_fixture(name)
    chk = stages.StageChecks(orc, ref, ref_inputs.CASES[name](), _one_iteration(name), not name.startswith("keyframes"))
    try:
        chk.run(stage)
    except AssertionError as e:
        raise AssertionError(f"stage '{stage}' of {name} differs from the reference -- decides: {stages.HINT[stage]}\n{e}") from e
    if stage in chk.report:
        print(f"[ref stage] {name} {stage}: {chk.report[stage]}")


@pytest.mark.parametrize("name", list(ref_inputs.CASES))
def test_oracle_poses_after_one_iteration(orc, name):
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
    ("TRANSFORM_PAIRWISE", ["global_points"], ["table", "member_lists", "info_mats", "residuals", "normal_equations"]),
    ("SUM3_LEFT", ["residuals"], ["table", "global_points", "member_lists", "normal_equations"]),
    ("MAHA_ASSOC", ["residuals"], ["table", "global_points", "member_lists", "info_mats", "normal_equations"]),
    ("FIT_FLOAT", ["info_mats"], ["table", "global_points", "member_lists", "residuals", "normal_equations"]),
    ("FIT_COV_GEMM", [], ["table", "global_points", "member_lists", "info_mats", "residuals", "normal_equations"]),
])
def test_every_float_order_hypothesis_is_decided_by_one_stage(orc, tmp_path, variant, fails, passes):
    """A dump written by the oracle built with ONE alternative reading stands in for "the reference turned out to evaluate it the other
    way": exactly the stage that owns the statement fails (stages condition on the reference's result of the stage before), the stages
    before and the independent ones pass.  FIT_FLOAT fails through the weights (their mean as a scalar chain is 11 ulp away from Eigen's
    redux order); FIT_COV_GEMM -- the one order that depends on the reference machine's cache sizes -- stays inside the stated bound of the
    information matrices but leaves hardly any matrix bit-equal: the report line of that stage is what tells those readings apart."""
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
        assert chk.report["info_mats"]["weights_max_ulp"] > 2 and chk.report["residuals"]["rows_differing"] == 0
    if variant == "FIT_COV_GEMM":
        assert chk.report["info_mats"]["matrices_bit_equal"] < 0.5 and chk.report["info_mats"]["weights_max_ulp"] == 0, chk.report["info_mats"]


def test_ref_inputs_are_deterministic(tmp_path):
    """the dumps the reference harness reads are a pure function of the committed seeds"""
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    dump.write_window_problem(a, ref_inputs.CASES["window_small"]())
    dump.write_window_problem(b, ref_inputs.CASES["window_small"]())
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a, "rb").read(8) == b"DMSAWN01"
