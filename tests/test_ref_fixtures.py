"""Parity against the REAL reference.  scripts/build_ref_oracle.sh (needs Eigen 3.4 / PCL 1.10 / Boost / ROS headers -- absent from the
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


def test_ref_inputs_are_deterministic(tmp_path):
    """the dumps the reference harness reads are a pure function of the committed seeds"""
    a, b = str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    dump.write_window_problem(a, ref_inputs.CASES["window_small"]())
    dump.write_window_problem(b, ref_inputs.CASES["window_small"]())
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a, "rb").read(8) == b"DMSAWN01"
