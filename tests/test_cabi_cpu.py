"""CPU-side checks of the product boundary: the C-ABI library builds/loads, exports every symbol declared in
include/dmsa_hip.h, and refuses to run without a GPU (no compute calls here)."""
import ctypes as C

import numpy as np
import os
import re

import pytest

from dmsa_lidar_slam_amd import _capi as capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(capi.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return capi.load_library()


def test_exports_every_declared_symbol(lib):
    header = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("dmsa_hip.h", "dmsa_debug.h", "dmsa_window_ring.h", "dmsa_static_points.h", "dmsa_window_setup.h", "dmsa_wire_formats.h", "dmsa_raw_sequence.h", "dmsa_keyframe_cloud.h", "dmsa_keyframe_map.h", "dmsa_aos.h"))
    declared = set(re.findall(r"\b(dmsa_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(capi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header_layout():
    # natural-alignment layouts of the PODs in include/dmsa_hip.h (x86-64 SysV)
    assert C.sizeof(capi.DebugOptions) == 33 * 4  # include/dmsa_debug.h: thirty-three int32 fields (append-only since round 6)
    assert C.sizeof(capi.Settings) == 72
    assert C.sizeof(capi.Report) == 48
    assert C.sizeof(capi.VoxelLevelInfo) == 56
    assert C.sizeof(capi.Timing) == 80
    assert C.sizeof(capi.WindowProblem) == 192
    assert C.sizeof(capi.KeyframeProblem) == 360
    assert C.sizeof(capi.StaticSelectProblem) == 80 and C.sizeof(capi.StaticSelectResult) == 24  # dmsa_static_points.h
    assert C.sizeof(capi.PreprocessConfig) == 80
    assert C.sizeof(capi.TrajState) == 120  # dmsa_window_setup.h
    assert C.sizeof(capi.PointCloud2) == 56  # dmsa_wire_formats.h
    assert C.sizeof(capi.AosView) == 40  # dmsa_aos.h
    assert C.sizeof(capi.DebugCounters) == 112  # dmsa_debug.h: fourteen int64 counters (append-only since round 6)


def test_default_settings_match_reference_defaults(lib):
    s = capi.Settings()
    lib.dmsa_default_settings(C.byref(s))
    # DmsaOptimizer.h:25-39
    assert (s.num_iter, s.epsilon, s.step_length_optim, s.max_step) == (15, 1e-5, 0.05, 0.01)
    assert (s.gauss_split, s.min_num_points_per_set, s.min_num_gaussians, s.use_centralization) == (0, 6, 30, 1)
    assert abs(s.grid_size_1_factor - 2.0) < 1e-7 and abs(s.grid_size_2_factor - 5.0) < 1e-7 and abs(s.lambda_diag - 1e-5) < 1e-12


def test_no_gpu_means_loud_failure_not_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ctx = C.c_void_p()
    assert lib.dmsa_create(0, 0, C.byref(ctx)) == capi.DMSA_ERR_NO_DEVICE
    from dmsa_lidar_slam_amd.api import DmsaError, DmsaOptimizer

    with pytest.raises(DmsaError):
        DmsaOptimizer()


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under dmsa_lidar_slam_amd/ may load, link or include it."""
    pkg = os.path.join(ROOT, "dmsa_lidar_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_py" not in txt and "dmsa_oracle" not in txt and "libdmsa_oracle" not in txt, f


@pytest.mark.parametrize("P", [12, 30, 63, 64, 65, 96, 186])
def test_host_lm_solve_equals_the_oracle_step_for_any_thread_count(lib, orc, P):
    """DmsaOptimizer.h:110-113 on the host side of the product: the explicit inverse by Gauss-Jordan, rows of a pivot step spread
    over worker threads for P >= 64 (the keyframe pass: 60 % of an iteration before).  Bit-identical to the oracle's lm_step and
    independent of the thread count."""


    rng = np.random.default_rng(100 + P)
    rows = 4 * P + 50
    e0 = rng.uniform(0.5, 2.0, rows)
    eb = e0[None, :] + rng.normal(size=(P, rows)) * 1e-4
    # a few exactly dependent columns exercise the pivot swaps under the weak damping of the reference (lambda = 1e-5)
    eb[P // 3] = eb[P // 3 + 1]
    h = float(np.sqrt(np.finfo(np.float32).eps))
    H, g, step_ref = orc.lm_step(e0, eb, h, float(np.float32(1e-5)), 0.2)   # H comes back damped
    for threads in (1, 3, 8):
        step = np.zeros(P)
        assert lib.dmsa_lm_solve(capi.ptr(H, C.c_double), capi.ptr(g, C.c_double), P, 0.2, threads, capi.ptr(step, C.c_double)) == capi.DMSA_OK
        assert np.array_equal(step, step_ref), (P, threads, np.abs(step - step_ref).max())
    assert lib.dmsa_lm_solve(None, capi.ptr(g, C.c_double), P, 0.2, 1, capi.ptr(step_ref, C.c_double)) == capi.DMSA_ERR_INVALID
