"""Behaviour of the C ABI around the happy path: context reuse across problems of different size and model, several
contexts at once, invalid arguments (negative return codes, never a crash), determinism of repeated calls."""
import ctypes as C

import numpy as np
import pytest

from dmsa_lidar_slam_amd import _capi as capi
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

pytestmark = pytest.mark.gpu


def _poses(p):
    return np.concatenate([p.relOrientations.ravel(), p.relTranslations.ravel()])


def test_context_reuse_across_sizes_and_models(hip, orc):
    """One context: small window, larger window, keyframe set, small window again -- buffers regrow, results stay those of a
    fresh context (bit for bit on the parity path)."""
    s_w, s_k = DmsaOptimSettings.sliding_window(num_iter=2), DmsaOptimSettings.keyframe_map(num_iter=2)
    probs = [(synth.window_problem(seed=31, scans=2, rings=16, az_steps=128, num_static=2000), s_w),
             (synth.window_problem(seed=32, scans=4, rings=32, az_steps=256, num_static=9000), s_w),
             (synth.keyframe_problem(seed=33, frames=5, rings=16, az_steps=128, arc=0.3), s_k),
             (synth.window_problem(seed=31, scans=2, rings=16, az_steps=128, num_static=2000), s_w)]
    shared = hip.DmsaOptimizer()
    got = []
    for p, s in probs:
        q = p.copy()
        shared.optimizeSet(q, s)
        got.append(_poses(q))
        fresh = p.copy()
        hip.DmsaOptimizer().optimizeSet(fresh, s)
        assert np.array_equal(got[-1], _poses(fresh))
    assert np.array_equal(got[0], got[3])


def test_two_contexts_interleaved(hip):
    s = DmsaOptimSettings.sliding_window(num_iter=2)
    pa = synth.window_problem(seed=41, scans=2, rings=16, az_steps=128, num_static=2000)
    pb = synth.window_problem(seed=42, scans=3, rings=16, az_steps=128, num_static=1000)
    a, b = hip.DmsaOptimizer(), hip.DmsaOptimizer()
    a.upload(pa), b.upload(pb)
    ra1, rb1 = a.optimizeResident(s), b.optimizeResident(s)
    xa, xb = a.poses(), b.poses()
    a2, b2 = hip.DmsaOptimizer(), hip.DmsaOptimizer()
    b2.upload(pb), a2.upload(pa)
    b2.optimizeResident(s), a2.optimizeResident(s)
    assert np.array_equal(xa[0], a2.poses()[0]) and np.array_equal(xb[1], b2.poses()[1])  # deterministic, no cross-talk
    assert (ra1.num_gaussians, rb1.num_gaussians) != (0, 0)


def test_the_path_is_deterministic(hip):
    p = synth.window_problem(seed=43, scans=3, rings=32, az_steps=192, num_static=4000)
    s = DmsaOptimSettings.sliding_window(num_iter=3)
    a, b = p.copy(), p.copy()
    hip.DmsaOptimizer().optimizeSet(a, s)
    hip.DmsaOptimizer().optimizeSet(b, s)
    assert np.array_equal(_poses(a), _poses(b))


def test_invalid_arguments_return_error_codes():
    lib = capi.load_library()
    ctx = C.c_void_p()
    assert lib.dmsa_create(10_000, 0, C.byref(ctx)) == capi.DMSA_ERR_NO_DEVICE      # no such device
    assert lib.dmsa_create(0, 0, None) == capi.DMSA_ERR_INVALID
    assert lib.dmsa_create(0, 0x10, C.byref(ctx)) == capi.DMSA_ERR_INVALID     # an unknown flag bit (0x10: the retired fast sums)
    assert lib.dmsa_create(0, 0, C.byref(ctx)) == capi.DMSA_OK
    rep, st = capi.Report(), capi.Settings()
    lib.dmsa_default_settings(C.byref(st))
    assert (st.num_iter, st.min_num_gaussians, st.min_num_points_per_set) == (15, 30, 6)       # DmsaOptimizer.h:25-39
    assert lib.dmsa_optimize_resident(ctx, C.byref(st), C.byref(rep)) == capi.DMSA_ERR_INVALID  # nothing uploaded
    assert lib.dmsa_optimize_window(ctx, None, C.byref(st), C.byref(rep)) == capi.DMSA_ERR_INVALID
    n = C.c_int32(0)
    assert lib.dmsa_num_table_rows(ctx, C.byref(n)) == capi.DMSA_ERR_INVALID
    wp = capi.WindowProblem()  # all-zero problem: no control poses
    assert lib.dmsa_window_upload(ctx, C.byref(wp)) == capi.DMSA_ERR_INVALID
    flags = (C.c_uint8 * 4)()
    assert lib.dmsa_radius_exists(ctx, None, 5, None, 0, 0.3, flags) == capi.DMSA_ERR_INVALID
    assert lib.dmsa_radius_exists(ctx, None, 0, None, 0, -1.0, flags) == capi.DMSA_ERR_INVALID
    assert isinstance(lib.dmsa_last_error(ctx), bytes)
    lib.dmsa_destroy(ctx)
    lib.dmsa_destroy(None)  # harmless


def test_coincident_control_stamps_are_rejected(hip):
    """boost::math::barycentric_rational throws std::logic_error on coincident nodes (uncaught in the reference); the ABI
    reports DMSA_ERR_INVALID instead."""
    p = synth.window_problem(seed=44, scans=2, rings=16, az_steps=96, num_static=500)
    p.stamps = p.stamps.copy()
    p.stamps[2] = p.stamps[1]
    with pytest.raises(hip.DmsaError):
        hip.DmsaOptimizer().optimizeSet(p, DmsaOptimSettings.sliding_window(num_iter=1))


def test_malformed_problems_are_rejected(hip):
    """What the reference would turn into NaN tables or out-of-bounds reads comes back as DMSA_ERR_INVALID: two control poses
    (Floater-Hormann d = 2 needs three), IMU parameter indices outside the time grid, non-positive minGridSize, frame offsets
    that decrease, a null normal array."""
    s = DmsaOptimSettings.sliding_window(num_iter=1)
    two = synth.window_problem(seed=38, scans=2, rings=8, az_steps=64, num_static=100, num_control_poses=2)
    with pytest.raises(hip.DmsaError):
        hip.DmsaOptimizer().optimizeSet(two, s)
    imu = synth.window_problem(seed=39, scans=2, rings=8, az_steps=64, num_static=100, use_imu=True)
    imu.paramIndices = imu.paramIndices.copy()
    imu.paramIndices[-1] = imu.trajTime.shape[0] + 5
    with pytest.raises(hip.DmsaError):
        hip.DmsaOptimizer().optimizeSet(imu, DmsaOptimSettings.sliding_window(use_imu=True, num_iter=1))
    grid = synth.window_problem(seed=40, scans=2, rings=8, az_steps=64, num_static=100)
    grid.minGridSize = 0.0
    with pytest.raises(hip.DmsaError):
        hip.DmsaOptimizer().optimizeSet(grid, s)
    kf = synth.keyframe_problem(seed=41, frames=3, rings=8, az_steps=64, arc=0.1)
    kf.frameOffsets = kf.frameOffsets.copy()
    kf.frameOffsets[1], kf.frameOffsets[2] = kf.frameOffsets[2], kf.frameOffsets[1]
    with pytest.raises(hip.DmsaError):
        hip.DmsaOptimizer().optimizeSet(kf, DmsaOptimSettings.keyframe_map(num_iter=1))


def test_repeated_centralize_keeps_the_origin(hip):
    """dmsa_centralize twice in a row must not lose the window origin (the second call is a no-op), and decentralize restores it."""
    p = synth.window_problem(seed=42, scans=2, rings=8, az_steps=64, num_static=500)
    opt = hip.DmsaOptimizer()
    opt.upload(p)
    before = opt.poses()
    opt.centralize(), opt.centralize()
    opt.decentralize(), opt.decentralize()
    after = opt.poses()
    assert np.abs(before[1] - after[1]).max() < 1e-12 and np.abs(before[0] - after[0]).max() < 1e-12


def test_create_destroy_cycles_release_device_memory(hip):
    """45 contexts created, used and destroyed in a row: after the first few (the HIP runtime keeps ~200 MB of code objects /
    pools once every kernel has run) the free device memory no longer moves."""
    rt = C.CDLL("libamdhip64.so")  # the runtime the library itself is linked against
    p = synth.window_problem(seed=45, scans=2, rings=16, az_steps=128, num_static=2000)
    s = DmsaOptimSettings.sliding_window(num_iter=1)

    def cycles(k):
        for _ in range(k):
            o = hip.DmsaOptimizer()
            o.optimizeSet(p.copy(), s)
            o.close()
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert rt.hipDeviceSynchronize() == 0 and rt.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    import gc

    gc.collect()
    free_a = cycles(15)
    free_b = cycles(30)
    assert free_a - free_b < 16 * 1024 * 1024, (free_a, free_b)  # one-sided: contexts of earlier tests may be collected meanwhile
