"""The header-only C++ class surface (include/dmsa_hip.hpp: DmsaOptimizer<PointT>::optimizeSet on ContinuousTrajectory /
MapManagement with the reference's member names) compiled with plain g++ against libdmsa_hip.so.

CPU: it compiles, links and refuses to run without a GPU (no CPU fallback).  GPU: same poses as the Python mirror of the
interface (bit for bit: both end in the same C calls) and within 1e-4 of the oracle on the parity path."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dmsa_lidar_slam_amd", "csrc")


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    from dmsa_lidar_slam_amd import _capi

    _capi.load_library()  # builds the library if needed and fails loudly if it cannot
    exe = str(tmp_path_factory.mktemp("cpp") / "wrapper_demo")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "wrapper_demo.cpp"),
           "-L", CSRC, "-ldmsa_hip", f"-Wl,-rpath,{CSRC}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def _dump(prob, settings, d):
    f = lambda a, t: np.ascontiguousarray(a, dtype=t)
    f(prob.relOrientations, np.float64).tofile(os.path.join(d, "ro.f64"))   # (C,3) rows == 3xC column-major
    f(prob.relTranslations, np.float64).tofile(os.path.join(d, "rt.f64"))
    f(prob.stamps, np.float64).tofile(os.path.join(d, "stamps.f64"))
    f(prob.trajTime, np.float64).tofile(os.path.join(d, "trajtime.f64"))
    f(prob.localPoints, np.float32).tofile(os.path.join(d, "local.f32"))
    f(prob.tformIdPerPoint, np.int32).tofile(os.path.join(d, "tidx.i32"))
    f(prob.ringIds, np.int32).tofile(os.path.join(d, "ring.i32"))
    f(prob.staticPoints, np.float32).tofile(os.path.join(d, "static.f32"))
    f(prob.staticRingIds, np.int32).tofile(os.path.join(d, "sring.i32"))
    np.array([prob.minGridSize, settings.num_iter, settings.step_length_optim, settings.max_step, settings.min_num_points_per_set],
             np.float64).tofile(os.path.join(d, "meta.f64"))


def _small():
    from dmsa_lidar_slam_amd import synth
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

    return synth.window_problem(seed=21, scans=3, rings=32, az_steps=192, num_static=4000), DmsaOptimSettings.sliding_window(num_iter=3)


def test_cpp_wrapper_compiles_and_refuses_cpu(demo, tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    prob, s = _small()
    _dump(prob, s, str(tmp_path))
    r = subprocess.run([demo, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "no usable HIP device" in r.stderr
    assert not os.path.exists(tmp_path / "out_ro.f64")


@pytest.mark.gpu
def test_cpp_wrapper_matches_python_mirror_and_oracle(demo, tmp_path, hip, orc):
    from dmsa_lidar_slam_amd import _capi

    prob, s = _small()
    _dump(prob, s, str(tmp_path))
    flags = _capi.FLAG_MIRROR_SUMS | _capi.FLAG_POSE_TABLE_HOST
    r = subprocess.run([demo, str(tmp_path), str(flags)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ro = np.fromfile(tmp_path / "out_ro.f64").reshape(-1, 3)
    rt = np.fromfile(tmp_path / "out_rt.f64").reshape(-1, 3)
    gl = np.fromfile(tmp_path / "out_global.f32", dtype=np.float32).reshape(-1, 4)
    p_py, p_ref = prob.copy(), prob.copy()
    opt = hip.DmsaOptimizer()
    rep = opt.optimizeSet(p_py, s)
    assert f"iterations {rep.iterations} stop_reason {rep.stop_reason}" in r.stdout
    assert np.array_equal(ro, p_py.relOrientations) and np.array_equal(rt, p_py.relTranslations)
    assert np.array_equal(gl[:, :3], opt.globalPoints()[:, :3])
    orc.optimize_window(p_ref, s)
    assert np.abs(rt - p_ref.relTranslations).max() < 1e-4 and np.abs(ro - p_ref.relOrientations).max() < 1e-4
    assert np.abs(rt - prob.relTranslations).max() > 1e-4  # it moved
