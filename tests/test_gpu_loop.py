"""The device-resident optimizeSet loop (csrc/loop_kernels.hip): the LM step of DmsaOptimizer.h:107-128 computed on the device (one
workgroup for P <= 64, panel hand-over between column-block workgroups beyond) against the oracle's step bit for bit; and whole
optimizeSet calls that END EARLY (no improvement / epsilon / too few Gaussians) -- the stop decision is taken on the device and reaches
the host one synchronisation late -- against the oracle and against the host-driven loop of rounds 1-2 (debug switch device_loop = 0)."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

pytestmark = pytest.mark.gpu


def _system(orc, P, seed):
    rng = np.random.default_rng(seed)
    rows = 4 * P + 50
    e0 = rng.uniform(0.5, 2.0, rows)
    eb = e0[None, :] + rng.normal(size=(P, rows)) * 1e-4
    eb[P // 3] = eb[P // 3 + 1]  # exactly dependent columns: pivot swaps under the weak damping of the reference (lambda = 1e-5)
    h = float(np.sqrt(np.finfo(np.float32).eps))
    return orc.lm_step(e0, eb, h, float(np.float32(1e-5)), 0.2)  # H (damped), g, step


@pytest.mark.parametrize("P,lm_stream", [(P, 1) for P in (6, 12, 30, 31, 63, 64, 65, 72, 96, 127, 128, 129, 184, 186, 187, 192, 193, 200, 594)]
                         + [(P, 0) for P in (65, 72, 186, 192)])
def test_device_lm_step_equals_the_oracle_step(hip, orc, P, lm_stream):
    """P <= 64: one workgroup (k_loop_lm_step); 64 < P <= 192: the stream of pivot-step records (k_loop_lm_stream; lm_stream = 0 selects the
    column-block panels, k_loop_lm_panels, which also serve P > 192) -- every path the oracle's step bit for bit."""
    H, g, step_ref = _system(orc, P, 300 + P)
    opt = hip.DmsaOptimizer(debug={"lm_stream": lm_stream})
    for rep in range(2):  # the panel solve reuses its scratch with a new epoch
        step, nan = opt.lmSolveDevice(H.reshape(P, P), g, 0.2)
        assert not nan
        assert np.array_equal(step, step_ref), (P, rep, np.abs(step - step_ref).max())
    # the clamp of :125-128
    max_step = 0.37 * np.abs(step_ref).max()
    mx, mn = step_ref.max(), step_ref.min()
    max_elem = max(mx, -mn)
    step, nan = opt.lmSolveDevice(H.reshape(P, P), g, 0.2, max_step)
    assert not nan and np.array_equal(step, (max_step / max_elem) * step_ref)
    # a NaN in the system is reported, not clamped
    Hn = H.reshape(P, P).copy()
    Hn[P // 2, P // 2] = np.nan
    _, nan = opt.lmSolveDevice(Hn, g, 0.2)
    assert nan
    opt.close()


def _run(hip, prob, s, debug=None):
    p = prob.copy()
    opt = hip.DmsaOptimizer(debug=debug)
    rep = opt.optimizeSet(p, s)
    tr = opt.trace()
    opt.close()
    return p, rep, tr


def _same(a, b):
    (pa, ra, ta), (pb, rb, tb) = a, b
    assert (ra.iterations, ra.stop_reason, ra.evaluations, ra.num_gaussians, ra.num_gaussians_l1, ra.num_memberships) == \
           (rb.iterations, rb.stop_reason, rb.evaluations, rb.num_gaussians, rb.num_gaussians_l1, rb.num_memberships)
    assert (ra.error0, ra.last_step_norm, ra.last_line_search_k) == (rb.error0, rb.last_step_norm, rb.last_line_search_k)
    assert [(t["M"], t["M1"], t["Mm"], t["best_k"], t["error0"], t["step_norm"]) for t in ta[: ra.iterations]] == \
           [(t["M"], t["M1"], t["Mm"], t["best_k"], t["error0"], t["step_norm"]) for t in tb[: rb.iterations]]
    assert np.array_equal(pa.relOrientations, pb.relOrientations) and np.array_equal(pa.relTranslations, pb.relTranslations)


def _cases():
    kf = synth.keyframe_problem(seed=5, frames=6, rings=24, az_steps=160, arc=0.4)
    rng = np.random.default_rng(0)
    ro, rt = kf.truth_relative
    kf.useOdometryErrorTerms = True
    kf.odomRelTransl = rt + rng.normal(0, 0.005, rt.shape)
    kf.odomRelOrientMat = (Rot.from_rotvec(ro) * Rot.from_rotvec(rng.normal(0, 1e-3, (kf.numFrames, 3)))).as_matrix()
    kf.__post_init__()
    return {
        "window_runs_to_a_stop": (synth.window_problem(seed=31, scans=3, rings=32, az_steps=256, num_static=4000), DmsaOptimSettings.sliding_window(num_iter=15), True),
        "window_imu": (synth.window_problem(seed=7, scans=5, rings=16, az_steps=256, num_static=1500, use_imu=True),
                                      DmsaOptimSettings.sliding_window(use_imu=True, num_iter=15), True),
        "rosette": (synth.rosette_window_problem(seed=2, scans=4, pts_per_scan=6000, num_static=3000), DmsaOptimSettings.sliding_window(num_iter=12), True),
        "keyframes_gravity_odometry": (kf, DmsaOptimSettings.keyframe_map(num_iter=12), False),
        "keyframes_P72": (synth.keyframe_problem(seed=4, frames=13, rings=16, az_steps=96, arc=0.8), DmsaOptimSettings.keyframe_map(num_iter=6), False),
    }


@pytest.mark.parametrize("case", list(_cases()))
def test_early_exits_match_the_oracle_and_the_host_driven_loop(hip, orc, case):
    prob, s, window = _cases()[case]
    dev = _run(hip, prob, s)
    p_ref = prob.copy()
    rep_ref, _, tr_ref = (orc.optimize_window if window else orc.optimize_keyframes)(p_ref, s)
    _same(dev, (p_ref, rep_ref, tr_ref))
    host = _run(hip, prob, s, debug={"device_loop": 0})
    _same(dev, host)
    # the stream dependencies as HIP events instead of device counters (csrc/dev_sync.h), and the single-workgroup slot scan
    _same(dev, _run(hip, prob, s, debug={"device_sync": 0, "fused_leaf_scan": 0}))
    _same(dev, _run(hip, prob, s, debug={"device_loop": 0, "device_sync": 0}))
    _same(dev, _run(hip, prob, s, debug={"shared_rotations": 0}))  # every evaluation of the Jacobian batch transforms its members itself
    if case == "window_imu":  # the trial chains' IMU rows in ONE kernel in front of the trial batch instead of on the side stream beside it
        _same(dev, _run(hip, prob, s, debug={"trial_rows_aside": 0}))
    # where the lane-per-evaluation tier ends and the workgroup tiers begin (the default follows the problem size: 32 members for these
    # small sets, 256 for large ones) changes nothing: every tier of the correspondence kernels and of the fit computes the same bits
    for thr in (256, 8, 100):
        _same(dev, _run(hip, prob, s, debug={"small_threshold": thr}))
    if case == "window_runs_to_a_stop":
        assert dev[1].stop_reason != 0 and dev[1].iterations < s.num_iter  # the case really exercises a device-side stop


def test_too_few_gaussians_leaves_the_state_of_the_iteration_start(hip, orc):
    """numPointSets < min_num_gaussians (DmsaOptimizer.h:89-93) in the SECOND iteration: the Jacobian batch of that iteration was already
    enqueued beside the voxelisation; nothing of it may reach the poses."""
    prob = synth.window_problem(seed=11, scans=2, rings=16, az_steps=128, num_static=800)
    s = DmsaOptimSettings.sliding_window(num_iter=4)
    p0 = prob.copy()
    rep0, _, tr0 = orc.optimize_window(p0, s)
    s.min_num_gaussians = tr0[1]["M"] + 1 if rep0.iterations > 1 else tr0[0]["M"] + 1  # iteration 1 passes only if it has more sets than iteration 2
    if rep0.iterations > 1 and tr0[0]["M"] <= tr0[1]["M"]:
        s.min_num_gaussians = tr0[0]["M"] + 1  # falls back to stopping in the first iteration
    p_ref = prob.copy()
    rep_ref, _, tr_ref = orc.optimize_window(p_ref, s)
    assert rep_ref.stop_reason == 1
    _same(_run(hip, prob, s), (p_ref, rep_ref, tr_ref))


def test_many_contexts_in_one_process_share_hardware_queues(hip, orc):
    """The device-side stream dependencies (csrc/dev_sync.h) must hold when HIP maps the streams of several contexts onto the same
    hardware queues, and when a context reuses the allocation (and so the stale counters) of one that was closed: eight contexts are
    created, used alternately and closed in a different order.  A wait that gave up would surface as DMSA_ERR_HIP."""
    from dmsa_lidar_slam_amd.api import DmsaOptimizer

    prob = synth.window_problem(seed=3, scans=3, rings=32, az_steps=256, num_static=3000)
    s = DmsaOptimSettings.sliding_window(num_iter=4)
    p_ref = prob.copy()
    orc.optimize_window(p_ref, s)
    opts = [DmsaOptimizer(device=0) for _ in range(8)]
    results = []
    for rounds in range(2):
        for o in opts:
            p = prob.copy()
            o.optimizeSet(p, s)
            results.append(p)
        opts[rounds].close()
        opts[rounds] = DmsaOptimizer(device=0)  # lands on the memory the closed context just freed
    for o in opts:
        o.close()
    for p in results:
        assert np.array_equal(p.relOrientations, p_ref.relOrientations) and np.array_equal(p.relTranslations, p_ref.relTranslations)


def _run_keep(hip, prob, s, debug=None):
    p = prob.copy()
    opt = hip.DmsaOptimizer(debug=debug)
    rep = opt.optimizeSet(p, s)
    return (p, rep, opt.trace()), opt


def test_pairs_equal_to_evaluation_0_are_left_out_of_the_jacobian_batch(hip, orc):
    """Perturbing relative pose k of the keyframe chain leaves the frames in front of k bit-identical (ConsecutivePoses.h:26-43,
    MapManagement.h:197-202), so a Gaussian whose members all lie in earlier frames has the residual of evaluation 0 in that evaluation.
    Default (eval_skip = 1): such (Gaussian, evaluation) pairs are not computed.  eval_skip = 0 computes every pair, eval_skip = 2 computes
    every pair AND compares the ones 1 would have left out with evaluation 0 bit for bit.  Same poses, same traces, zero mismatches -- at
    P = 72 and at the P = 186 of the bench's neighbourhoods, through the matrix-core normal equations."""
    for frames, iters in ((13, 3), (32, 2)):
        prob = synth.keyframe_problem(seed=4, frames=frames, rings=16, az_steps=96, arc=0.8 if frames == 13 else 2.0)
        s = DmsaOptimSettings.keyframe_map(num_iter=iters)
        p_ref = prob.copy()
        rep_ref, _, tr_ref = orc.optimize_keyframes(p_ref, s)
        skip, o1 = _run_keep(hip, prob, s, debug={"skip_stats": 1})
        full, o0 = _run_keep(hip, prob, s, debug={"eval_skip": 0, "skip_stats": 1})
        both, o2 = _run_keep(hip, prob, s, debug={"eval_skip": 2, "skip_stats": 1})
        _same(skip, (p_ref, rep_ref, tr_ref))
        _same(skip, full)
        _same(skip, both)
        c1, c0, c2 = o1.debugCounters(), o0.debugCounters(), o2.debugCounters()
        for o in (o0, o1, o2):
            o.close()
        P = 6 * (frames - 1)
        assert c0["skip_pairs"] == 0  # switched off: nobody looks
        assert c1["skip_pairs"] == c2["skip_pairs"] == sum(t["M"] for t in tr_ref[: rep_ref.iterations]) * P
        assert c1["skip_pairs_equal"] == c2["skip_pairs_equal"] > 0.1 * c1["skip_pairs"], c1  # a real share of the batch (these frames overlap a lot)
        assert c2["skip_mismatches"] == 0 and c1["skip_mismatches"] == 0
        print(f"[eval_skip] frames={frames} P={P}: {c1['skip_pairs_equal']} of {c1['skip_pairs']} pairs left out ({100.0 * c1['skip_pairs_equal'] / c1['skip_pairs']:.1f} %)")


def test_a_timed_out_device_side_wait_restarts_the_call_with_events(hip, orc):
    """Debug switch sync_fault = k withholds the signal of the k-th device-side wait of a call (csrc/dev_sync.h): the wait gives up, its
    consumer runs before its producer may have finished, nothing of the run can be trusted.  The library restores the state the call
    started from (poses AND the static points, which centralize / decentralize does not round-trip exactly), switches the context to
    event dependencies and runs the call again: same result as an undisturbed run, DMSA_OK with a warning, and the context keeps working."""
    prob = synth.window_problem(seed=31, scans=3, rings=32, az_steps=256, num_static=4000)
    s = DmsaOptimSettings.sliding_window(num_iter=5)
    clean = _run(hip, prob, s)
    for k in (1, 4, 9):
        got, opt = _run_keep(hip, prob, s, debug={"sync_fault": k})
        _same(clean, got)
        c = opt.debugCounters()
        assert c["sync_retries"] == 1, c
        assert "warning" in opt.lastError() and "timed out" in opt.lastError()
        p2 = prob.copy()  # the same context again: event dependencies now, no further retry
        rep2 = opt.optimizeSet(p2, s)
        _same(clean, (p2, rep2, opt.trace()))
        assert opt.debugCounters()["sync_retries"] == 1
        opt.close()
    kf = synth.keyframe_problem(seed=4, frames=13, rings=16, az_steps=96, arc=0.8)
    sk = DmsaOptimSettings.keyframe_map(num_iter=3)
    got, opt = _run_keep(hip, kf, sk, debug={"sync_fault": 3})
    _same(_run(hip, kf, sk), got)
    assert opt.debugCounters()["sync_retries"] == 1
    opt.close()


def test_a_wrong_sort_width_guess_reruns_the_voxelisation(hip, orc):
    """The radix sort of the voxel keys is sized from the PREVIOUS iteration's tree depth (csrc/voxelize_driver.cpp); a guess that turns
    out too small is seen with the counts and the voxelisation runs again, synchronously.  Debug switch speculation_fault = k plants a
    depth one too small in the k-th voxelisation of a call: same result, one retry counted; the cost of the branch is printed."""
    import time

    prob = synth.window_problem(seed=31, scans=3, rings=32, az_steps=256, num_static=4000)
    s = DmsaOptimSettings.sliding_window(num_iter=5)
    clean = _run(hip, prob, s)
    general = {"small_voxel": 0}  # (28 576 points: the one-launch path of small_voxel.hip has no sort width to guess -- this is about the general path)

    def timed(debug):
        opt = hip.DmsaOptimizer(fixed_iters=True, debug=dict(general, **(debug or {})))
        best = 1e9
        for _ in range(3):
            p = prob.copy()
            opt.upload(p)
            t0 = time.perf_counter()
            opt.optimizeResident(s)
            best = min(best, time.perf_counter() - t0)
        c = opt.debugCounters()
        opt.close()
        return best, c

    for k in (2, 3):
        got, opt = _run_keep(hip, prob, s, debug=dict(general, speculation_fault=k))
        _same(clean, got)
        assert opt.debugCounters()["speculation_retries"] == 1
        opt.close()
    t_clean, c_clean = timed(None)
    t_fault, c_fault = timed({"speculation_fault": 3})
    assert c_clean["speculation_retries"] == 0 and c_fault["speculation_retries"] == 3
    print(f"[speculation] 5 iterations: {1e3 * t_clean:.3f} ms clean, {1e3 * t_fault:.3f} ms with one mis-speculated voxelisation "
          f"(+{1e3 * (t_fault - t_clean):.3f} ms per wrong guess)")


def test_optimize_pose_tables_optimize_on_one_context(hip, orc):
    """Regression (advisor, round 3): a pose-table call that regrows the pinned control-pose ring between two optimize calls used to free
    the pinned per-iteration result buffer of the device loop without forgetting it."""
    prob = synth.keyframe_problem(seed=4, frames=13, rings=16, az_steps=96, arc=0.8)
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    p_ref = prob.copy()
    rep_ref, _, tr_ref = orc.optimize_keyframes(p_ref, s)
    opt = hip.DmsaOptimizer()
    p = prob.copy()
    rep = opt.optimizeSet(p, s)
    _same((p, rep, opt.trace()), (p_ref, rep_ref, tr_ref))
    P = 6 * (prob.numFrames - 1)
    params = np.tile(np.concatenate([prob.relOrientations[1:].reshape(-1), prob.relTranslations[1:].reshape(-1)]), (1 + P, 1))
    opt.poseTables(params)  # 1 + P evaluations: larger than anything the ring has held so far
    for _ in range(2):
        p = prob.copy()
        rep = opt.optimizeSet(p, s)
        _same((p, rep, opt.trace()), (p_ref, rep_ref, tr_ref))
    opt.close()


def test_a_keyframe_set_whose_chain_state_outgrows_64_kb_of_lds(hip):
    """Regression (advisor, round 3): the chain kernels keep (30 n + 2 P + rows) doubles in dynamic LDS -- above ~185 frames that is more
    than the 64 KB a kernel gets without asking, and every launch failed.  Now the limit is raised (160 KB: ~470 frames) and larger sets go
    to the host-driven loop.  200 frames of a few hundred points: P = 1194, beyond the panel solve too (host solve inside the device loop).
    Checked against the host-driven loop (which the smaller cases above pin to the oracle; the oracle itself needs minutes at this P)."""
    prob = synth.keyframe_problem(seed=8, frames=200, rings=8, az_steps=48, arc=5.5)
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    dev = _run(hip, prob, s)
    assert dev[1].iterations == 2 and dev[1].num_gaussians > 1000
    _same(dev, _run(hip, prob, s, debug={"device_loop": 0}))


def test_the_two_device_lm_solves_give_the_same_keyframe_pass(hip, orc):
    """64 < P <= 192 (the keyframe pass): the stream of pivot-step records (k_loop_lm_stream, default) and the column-block panels
    (lm_stream = 0) inside whole optimizeSet calls -- same poses, same line-search decisions, iteration by iteration."""
    prob = synth.keyframe_problem(seed=4, frames=13, rings=16, az_steps=96, arc=0.8)  # P = 72
    s = DmsaOptimSettings.keyframe_map(num_iter=5)
    _same(_run(hip, prob, s, debug={"lm_stream": 1}), _run(hip, prob, s, debug={"lm_stream": 0}))
    big = synth.keyframe_problem(seed=9, frames=32, rings=16, az_steps=128, arc=1.2)  # P = 186: three workgroups of A columns
    s2 = DmsaOptimSettings.keyframe_map(num_iter=3)
    _same(_run(hip, big, s2, debug={"lm_stream": 1}), _run(hip, big, s2, debug={"lm_stream": 0}))


def test_lattice_events_of_the_previous_voxelisation_are_verified_instead_of_replayed(hip, orc):
    """k_lattice (PCL's sequential adoptBoundingBoxToPoint) first checks IN PARALLEL whether the growth events of the previous voxelisation
    of the context still hold for the moved points; only if not does it replay.  Same bits either way (lattice_hint = 0: always replay), the
    counters show that the later iterations of a call take the fast path, and a context that moves on to ANOTHER problem replays."""
    prob = synth.window_problem(seed=23, scans=3, rings=32, az_steps=256, num_static=4000)
    other = synth.rosette_window_problem(seed=3, scans=3, pts_per_scan=5000, num_static=2000)
    s = DmsaOptimSettings.sliding_window(num_iter=5)
    ref, ref_other = prob.copy(), other.copy()
    orc.optimize_window(ref, s)
    orc.optimize_window(ref_other, s)
    for hint in (1, 0):
        opt = hip.DmsaOptimizer(debug={"lattice_hint": hint})
        a, b, c = prob.copy(), other.copy(), prob.copy()
        opt.optimizeSet(a, s), opt.optimizeSet(b, s), opt.optimizeSet(c, s)  # the same context: prob, another problem, prob again
        assert np.array_equal(a.getPoseParameters(), ref.getPoseParameters()) and np.array_equal(c.getPoseParameters(), ref.getPoseParameters())
        assert np.array_equal(b.getPoseParameters(), ref_other.getPoseParameters())
        cnt = opt.debugCounters()
        if hint:
            assert cnt["lattice_hints_held"] >= 2 * 3 * 3 and cnt["lattice_replays"] >= 2 * 3, cnt  # the first voxelisation of another problem replays
        else:
            assert cnt["lattice_hints_held"] == 0 and cnt["lattice_replays"] > 0
        opt.close()
