"""GPU parity on BASELINE.json's configurations and the edge cases of the path.

config 1 (one 16 384-point scan vs 50 k static, 5 iterations), config 3 (10 x 131 072 window, full size: structure and
residual parity against the oracle plus size-independent properties), config 4 (keyframe set with gravity AND odometry
rows, ragged frames), config 5 (rosette pattern, ids = k % 1000), and the degenerate inputs (no static map, non-finite
points through the whole optimizeSet, fewer frames than Gaussians need).
"""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

pytestmark = pytest.mark.gpu

H_INCR = float(np.sqrt(np.finfo(np.float32).eps))


def _pose_diff(orc, a, b):
    ga_o, ga_t = orc.relative2global(a.relOrientations, a.relTranslations)
    gb_o, gb_t = orc.relative2global(b.relOrientations, b.relTranslations)
    return np.abs(ga_t - gb_t).max(), np.abs(ga_o - gb_o).max()


def _parity_run(hip, orc, prob, s, window=True, debug=None):
    """Whole optimizeSet on the parity path against the oracle: same control flow, poses within 1e-4 m / 1e-4 rad."""
    p_ref, p_gpu = prob.copy(), prob.copy()
    fn = orc.optimize_window if window else orc.optimize_keyframes
    rep_ref, _, trace = fn(p_ref, s)
    opt = hip.DmsaOptimizer(debug=debug)
    rep = opt.optimizeSet(p_gpu, s)
    assert (rep.iterations, rep.stop_reason, rep.evaluations) == (rep_ref.iterations, rep_ref.stop_reason, rep_ref.evaluations)
    for a, b in zip(trace, opt.trace()):
        assert (a["M"], a["M1"], a["Mm"], a["best_k"]) == (b["M"], b["M1"], b["Mm"], b["best_k"])
        assert abs(a["error0"] - b["error0"]) <= 1e-9 * max(a["error0"], 1e-300)
    dt, dr = _pose_diff(orc, p_ref, p_gpu)
    assert dt < 1e-4 and dr < 1e-4, (dt, dr)
    return rep, p_gpu


# ---- config 1 ------------------------------------------------------------------------------------------------------
def test_config1_single_scan_vs_static_map(hip, orc):
    prob = synth.window_problem(seed=1, scans=1, rings=128, az_steps=128, num_static=50_000)
    assert prob.localPoints.shape[0] <= 16_384 and prob.staticPoints.shape[0] == 50_000
    rep, p_gpu = _parity_run(hip, orc, prob, DmsaOptimSettings.sliding_window(num_iter=5))
    assert rep.iterations >= 1
    moved_t, moved_r = _pose_diff(orc, prob, p_gpu)
    assert moved_t > 1e-4 or moved_r > 1e-4


# ---- config 5 ------------------------------------------------------------------------------------------------------
def test_config5_rosette_no_imu(hip, orc):
    prob = synth.rosette_window_problem(seed=2, scans=5, pts_per_scan=24_000, num_static=20_000)
    assert prob.ringIds.max() == 999  # ids = k % 1000: the ring-diversity test sees "rings" that are just indices
    _parity_run(hip, orc, prob, DmsaOptimSettings.sliding_window(num_iter=4))


# ---- config 4: additional rows, ragged frames ------------------------------------------------------------------------
def _with_odometry(prob, seed=0):
    rng = np.random.default_rng(seed)
    f = prob.numFrames
    ro, rt = prob.truth_relative
    prob.useOdometryErrorTerms = True
    prob.odomRelTransl = rt + rng.normal(0, 0.005, rt.shape)
    prob.odomRelOrientMat = (Rot.from_rotvec(ro) * Rot.from_rotvec(rng.normal(0, 1e-3, (f, 3)))).as_matrix()
    prob.__post_init__()
    return prob


def test_keyframes_gravity_and_odometry_rows(hip, orc):
    prob = _with_odometry(synth.keyframe_problem(seed=5, frames=6, rings=24, az_steps=160, arc=0.4))
    prob.gravityPlausible[2] = 0  # implausible frame: its row stays exactly zero (MapManagement.h:216-231)
    s = DmsaOptimSettings.keyframe_map(num_iter=3)
    _parity_run(hip, orc, prob, s, window=False)
    # the additional rows are really in the problem: switching them off changes the result
    plain = prob.copy()
    plain.useOdometryErrorTerms = False
    plain.useGravityErrorTerms = False
    a, b = prob.copy(), plain
    hip.DmsaOptimizer().optimizeSet(a, s)
    hip.DmsaOptimizer().optimizeSet(b, s)
    assert _pose_diff(orc, a, b)[0] > 1e-7


def test_keyframes_ragged_frames(hip, orc):
    """Frames of very different sizes (one of them nearly empty)."""
    prob = synth.keyframe_problem(seed=6, frames=6, rings=24, az_steps=160, arc=0.4)
    off = prob.frameOffsets
    keep = np.ones(prob.localPoints.shape[0], bool)
    keep[off[1] + 5:off[2]] = False            # frame 1 keeps 5 points
    keep[off[3]:off[4]][::2] = False           # frame 3 keeps every second point
    sizes = np.array([keep[off[k]:off[k + 1]].sum() for k in range(prob.numFrames)])
    prob.localPoints, prob.localNormals, prob.ringIds = prob.localPoints[keep], prob.localNormals[keep], prob.ringIds[keep]
    prob.frameOffsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    prob.__post_init__()
    _parity_run(hip, orc, prob, DmsaOptimSettings.keyframe_map(num_iter=2), window=False)


def test_split_search_skips_pair_blocks_that_cannot_be_antiparallel(hip):
    """splitSet's O(n^2) search (Gaussians.h:36-51) only matters for pairs with |n_a + n_c| <= 0.5 (:54).  k_split_pairs bounds every block of
    64 positions x 64 partners by the cone of the partners' normals and skips the blocks that cannot reach 0.5; the results are the ones the
    exhaustive oracle gives (every keyframe parity test runs through it) -- here: that the skipping really happens at the bench shape."""
    full = synth.keyframe_problem(seed=1, frames=32, arc=2 * np.pi * 32 / 256.0)
    opt = hip.DmsaOptimizer(debug={"skip_stats": 1})
    opt.optimizeSet(full.getSubmap(0, 31), DmsaOptimSettings.keyframe_map(num_iter=1))
    c = opt.debugCounters()
    opt.close()
    print(f"[splitSet] {c['split_blocks_skipped']} of {c['split_blocks']} 64 x 64 pair blocks skipped ({100.0 * c['split_blocks_skipped'] / c['split_blocks']:.1f} %)")
    assert c["split_blocks"] > 10_000 and c["split_blocks_skipped"] > 0.3 * c["split_blocks"]


def test_config4_at_the_shard_size_bench_times(hip, orc):
    """One neighbourhood exactly as `bench.py --workload keyframes` / the keyframe_pass key shards it: 32 keyframes x ~10^4 points
    (P = 186 parameters, 197 evaluations per iteration), gauss_split, gravity rows -- the default path against the oracle for two
    iterations: same Gaussian counts, same line-search decisions, poses within 1e-4 m / 1e-4 rad (the oracle needs ~1.5 s per
    iteration here)."""
    frames = 32
    full = synth.keyframe_problem(seed=1, frames=frames, arc=2 * np.pi * frames / 256.0)
    prob = full.getSubmap(0, frames - 1)
    assert prob.numParams == 186 and prob.localPoints.shape[0] > 300_000 and prob.useGravityErrorTerms
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    assert s.gauss_split
    rep, p_gpu = _parity_run(hip, orc, prob, s, window=False)
    assert rep.iterations == 2 and rep.evaluations == 2 * (186 + 10)
    moved_t, moved_r = _pose_diff(orc, prob, p_gpu)
    assert moved_t > 1e-5 or moved_r > 1e-5


def test_more_gaussians_than_points(hip, orc):
    """min_num_points_per_set = 0 with gauss_split: a leaf of three points (two ids on one face, a third point on the opposite face)
    splits into a set of two and a set of one, at both resolutions -- M = 4n/3 > n.  Buffers indexed by Gaussian (size-class order, fit
    sums) must hold that (they were sized n + 16); structure and one optimizeSet against the oracle."""
    from dmsa_lidar_slam_amd.problems import MapManagement

    rng = np.random.default_rng(3)
    cells, g = 400, 0.05
    centre = (rng.integers(-12, 12, (cells, 3)) * 40 * g + rng.uniform(-0.3 * g, 0.3 * g, (cells, 3))).astype(np.float32)
    centre = np.unique(centre.round(3), axis=0)
    cells = centre.shape[0]
    pts = np.repeat(centre, 3, axis=0) + rng.uniform(-0.05 * g, 0.05 * g, (3 * cells, 3)).astype(np.float32)
    nrm = np.tile(np.array([[0, 0, 1], [0, 0, 1], [0, 0, -1]], np.float32), (cells, 1))
    ids = np.tile(np.array([1, 2, 3], np.int32), cells)
    n = pts.shape[0]
    frames = 3
    off = np.array([0, n // 3, 2 * n // 3, n], np.int64)
    rel_o = np.zeros((frames, 3))
    rel_t = np.zeros((frames, 3))
    rel_t[1:] = rng.normal(0, 1e-3, (frames - 1, 3))
    prob = MapManagement(relOrientations=rel_o, relTranslations=rel_t, frameOffsets=off, localPoints=pts, localNormals=nrm, ringIds=ids, minGridSize=g)
    s = DmsaOptimSettings.keyframe_map(num_iter=2)
    s.min_num_points_per_set = 0
    p_ref = prob.copy()
    rep_ref, _, trace = orc.optimize_keyframes(p_ref, s)
    assert trace[0]["M"] > n + 16, (trace[0]["M"], n)
    rep, p_gpu = _parity_run(hip, orc, prob, s, window=False)
    assert rep.num_gaussians > n + 16


# ---- degenerate inputs ----------------------------------------------------------------------------------------------
def test_window_without_static_map(hip, orc):
    prob = synth.window_problem(seed=4, scans=3, rings=32, az_steps=256, num_static=0)
    assert prob.staticPoints.shape[0] == 0
    _parity_run(hip, orc, prob, DmsaOptimSettings.sliding_window(num_iter=3))


def test_window_with_nonfinite_points_end_to_end(hip, orc):
    prob = synth.window_problem(seed=12, scans=3, rings=32, az_steps=192, num_static=4000)
    prob.localPoints[0, 0] = np.nan
    prob.localPoints[777, 2] = np.inf
    prob.localPoints[5000:5016, 1] = np.nan
    prob.staticPoints[10, 0] = -np.inf
    rep, p_gpu = _parity_run(hip, orc, prob, DmsaOptimSettings.sliding_window(num_iter=3))
    assert np.isfinite(p_gpu.relTranslations).all() and np.isfinite(p_gpu.relOrientations).all()


def test_two_frame_keyframe_set_aborts_like_reference(hip, orc):
    prob = synth.keyframe_problem(seed=9, frames=2, rings=4, az_steps=24, arc=0.05)
    s = DmsaOptimSettings.keyframe_map(num_iter=3)
    p_ref, p_gpu = prob.copy(), prob.copy()
    rep_ref, _, _ = orc.optimize_keyframes(p_ref, s)
    rep = hip.DmsaOptimizer().optimizeSet(p_gpu, s)
    assert (rep.stop_reason, rep.iterations) == (rep_ref.stop_reason, rep_ref.iterations)
    assert _pose_diff(orc, p_ref, p_gpu)[0] < 1e-12


# ---- config 3 at full size ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full_window():
    return synth.window_problem(seed=1)  # 10 x 131 072 + 200 000 static, the bench workload


def test_config3_full_size_structure_and_residuals(hip, orc, full_window):
    """The bench workload itself: bit-exact voxel structure, Gaussian sets, information matrices and residuals; size-independent
    invariants."""
    prob = full_window
    s = DmsaOptimSettings.sliding_window()
    n, ns = prob.localPoints.shape[0], prob.staticPoints.shape[0]
    table, _ = orc.window_pose_table(prob)
    g = orc.transform_points(table, prob.localPoints, prob.tformIdPerPoint)
    glob = np.concatenate([g, prob.staticPoints]).astype(np.float32)
    ids = np.concatenate([prob.ringIds, prob.staticRingIds])
    ref = orc.Gaussians(glob, ids, prob.minGridSize, s)

    opt = hip.DmsaOptimizer()
    opt.upload(prob)
    opt.poseTables(prob.getPoseParameters())
    got = opt.updateGlobalPoints(0)
    assert np.array_equal(got[:n, :3], glob[:n, :3])
    M, Mm = opt.buildGaussians(s)
    assert (M, Mm) == (ref.M, ref.Mm)
    for level, f in ((0, s.grid_size_1_factor), (1, s.grid_size_2_factor)):
        info, code, key, order = opt.voxelLevel(level)
        res = float(np.float32(f) * np.float32(prob.minGridSize))
        info_r, code_r, key_r, order_r = orc.voxelize(glob, res)
        assert info.depth == info_r.depth and info.num_leaves == info_r.num_leaves
        assert np.array_equal(code, code_r) and np.array_equal(order, order_r)
        # invariants: DFS order = non-decreasing codes, `order` is a permutation of the finite points, ascending in a leaf
        sc = code[order]  # leaf codes are per point; `order` is the leaf-DFS permutation
        assert np.all(sc[1:] >= sc[:-1])
        assert np.array_equal(np.sort(order), np.arange(n + ns))
        same = sc[1:] == sc[:-1]
        assert np.all(order[1:][same] > order[:-1][same])
    seg, memb, info12, w = opt.gaussians()
    assert np.array_equal(seg, ref.seg_offset) and np.array_equal(memb, ref.members)
    assert seg[-1] == Mm and np.all(np.diff(seg) >= s.min_num_points_per_set)
    # the fit in Eigen's own float orders, incl. the Gaussians above 680 members whose products run in several depth blocks
    assert np.diff(seg).max() > 4096
    assert np.array_equal(info12, ref.info) and np.array_equal(w, ref.weights)
    base = prob.getPoseParameters()
    params = np.stack([base, base + H_INCR * np.eye(len(base))[4]])
    tables = opt.poseTables(params)
    e = opt.evalResiduals(2)
    for b in range(2):
        gb = orc.transform_points(tables[b], prob.localPoints, prob.tformIdPerPoint)
        e_ref = ref.residuals(np.concatenate([gb, prob.staticPoints]).astype(np.float32))
        assert np.array_equal(e[b], e_ref)
    # idempotence: building again from the same points gives the same sets
    assert opt.buildGaussians(s) == (M, Mm)
    seg2, memb2, _, _ = opt.gaussians()
    assert np.array_equal(seg2, seg) and np.array_equal(memb2, memb)
    opt.close()


def test_config3_full_size_two_iterations_match_oracle(hip, orc, full_window):
    rep, _ = _parity_run(hip, orc, full_window, DmsaOptimSettings.sliding_window(num_iter=2))
    assert rep.num_gaussians > 10_000


@pytest.mark.parametrize("dt_res", [2e-4, 2e-5])
def test_long_pose_tables(hip, orc, dt_res):
    """Dense pose tables of 10^3 .. 10^4 rows (fine dt_res)."""
    prob = synth.window_problem(seed=14, scans=2, rings=32, az_steps=256, num_static=3000, dt_res=dt_res)
    assert prob.trajTime.shape[0] > 900
    s = DmsaOptimSettings.sliding_window(num_iter=2)
    _parity_run(hip, orc, prob, s)


def test_additional_error_rows_bit_exact(orc):
    """getAdditionalErrorTerms through the C ABI: IMU rows of the window model (ContinuousTrajectory.h:603-663) and gravity rows of
    the keyframe model (MapManagement.h:210-232) are host double arithmetic in the reference's order: identical to the oracle."""
    from dmsa_lidar_slam_amd.api import DmsaOptimizer

    g = DmsaOptimizer(device=0)
    w = synth.window_problem(seed=3, scans=3, rings=16, az_steps=128, num_static=500, use_imu=True)
    g.upload(w)
    a, b = g.getAdditionalErrorTerms(), orc.window_additional_errors(w)
    assert a.shape == (5,) and np.array_equal(a, b) and np.all(a > 0)
    params = g.getPoseParameters()
    params[3] += 1e-3
    g.setPoseParameters(params)
    assert not np.array_equal(g.getAdditionalErrorTerms(), a)  # the rows follow the pose parameters
    k = synth.keyframe_problem(seed=5, frames=6, rings=16, az_steps=128, arc=0.4)
    g.upload(k)
    a, b = g.getAdditionalErrorTerms(), orc.keyframe_additional_errors(k)
    assert a.shape == b.shape and np.array_equal(a, b) and a[0] == 0.0  # row 0 stays exactly 0 (q13)
    w0 = synth.window_problem(seed=3, scans=3, rings=16, az_steps=128, num_static=500)
    g.upload(w0)
    assert g.getAdditionalErrorTerms().shape == (0,)
    g.close()


@pytest.mark.parametrize("case", ["window_static", "window_imu", "rosette", "keyframes_split", "keyframes_p72"])
def test_parity_path_trajectories_bit_identical(orc, case):
    """Serial-order sums, host pose tables and the shared reduction order of the normal equations: the parity path returns the
    oracle's poses bit for bit (not just within 1e-4), iteration counts, Gaussian counts and line-search decisions included."""
    from dmsa_lidar_slam_amd.api import DmsaOptimizer

    if case == "window_static":
        p, s, run = synth.window_problem(seed=6, scans=4, rings=32, az_steps=256, num_static=4000), DmsaOptimSettings.sliding_window(num_iter=4), orc.optimize_window
    elif case == "window_imu":
        p, s, run = synth.window_problem(seed=7, scans=5, rings=16, az_steps=256, num_static=1500, use_imu=True), DmsaOptimSettings.sliding_window(use_imu=True, num_iter=4), orc.optimize_window
    elif case == "rosette":
        p, s, run = synth.rosette_window_problem(seed=2, scans=4, pts_per_scan=6000, num_static=3000), DmsaOptimSettings.sliding_window(num_iter=4), orc.optimize_window
    elif case == "keyframes_split":
        p, s, run = synth.keyframe_problem(seed=3, frames=6, rings=16, az_steps=160, arc=0.4), DmsaOptimSettings.keyframe_map(num_iter=4), orc.optimize_keyframes
    else:  # 13 frames: P = 72 > 64 takes the other branch of the block-size rule
        p, s, run = synth.keyframe_problem(seed=4, frames=13, rings=16, az_steps=96, arc=0.8), DmsaOptimSettings.keyframe_map(num_iter=3), orc.optimize_keyframes
    a, b = p.copy(), p.copy()
    g = DmsaOptimizer(device=0)
    ra, tr_a = g.optimizeSet(a, s), g.trace()
    g.close()
    rb, _, tr_b = run(b, s)
    assert (ra.iterations, ra.stop_reason, ra.num_gaussians, ra.num_memberships, ra.evaluations) == (rb.iterations, rb.stop_reason, rb.num_gaussians, rb.num_memberships, rb.evaluations)
    assert [(t["M"], t["Mm"], t["best_k"], t["error0"], t["step_norm"]) for t in tr_a[: ra.iterations]] == [(t["M"], t["Mm"], t["best_k"], t["error0"], t["step_norm"]) for t in tr_b[: rb.iterations]]
    assert np.array_equal(a.relOrientations, b.relOrientations) and np.array_equal(a.relTranslations, b.relTranslations)
    assert ra.iterations >= 2


def test_parallel_second_pass_of_the_chain_tiers(hip, orc, full_window, monkeypatch):
    """The double sum of every Gaussian above 128 members is computed as a parallel reduction when integer bounds prove that the
    reference's member-by-member chain cannot round, otherwise by the chain.  Residuals must equal the oracle's
    sequential sums bit for bit in all three modes (parallel with test, parallel then chain anyway, chain only) -- also on a window
    with adversarial members: static points placed exactly on the float means of the largest Gaussians give terms that are zero or
    ~2^-40 beside sums ~2^6, for which the proof must fail and the chain run (counted by dmsa_serial_fallback_sums)."""
    s = DmsaOptimSettings.sliding_window()
    base = full_window
    table, _ = orc.window_pose_table(base)
    g = orc.transform_points(table, base.localPoints, base.tformIdPerPoint)
    glob = np.concatenate([g, base.staticPoints]).astype(np.float32)
    ids = np.concatenate([base.ringIds, base.staticRingIds])
    G = orc.Gaussians(glob, ids, base.minGridSize, s)
    sizes = np.diff(G.seg_offset)
    big = np.argsort(-sizes)[:6]
    assert sizes[big[0]] >= 4096
    extra = []
    for gi in big:
        P = glob[G.members[G.seg_offset[gi]:G.seg_offset[gi + 1]], :3]
        m = np.zeros(3, np.float32)
        for p in P:  # the reference's float chain
            m = (m + p).astype(np.float32)
        extra.append((m / np.float32(len(P))).astype(np.float32))
    adv = base.copy()
    pts = np.zeros((len(extra), adv.staticPoints.shape[1]), np.float32)
    pts[:, :3] = np.array(extra)
    adv.staticPoints = np.concatenate([adv.staticPoints, pts]).astype(np.float32)
    adv.staticRingIds = np.concatenate([adv.staticRingIds, np.full(len(extra), 7, adv.staticRingIds.dtype)])

    # ... and WINDOW points moved onto the mean of the other members of a large Gaussian, near the front, the middle and three quarters of its
    # member list: tiny terms in different blocks of the list, so that the block-wise fall-back of the latency tier (serial_kernels.hip) has to
    # chain blocks in the middle and carry a running sum with low bits into the blocks behind them
    adv_mid = base.copy()
    N = base.localPoints.shape[0]
    T = np.asarray(table, np.float64).reshape(-1, 3, 4)
    moved = 0
    for gi in big[:4]:
        mem = np.asarray(G.members[G.seg_offset[gi]:G.seg_offset[gi + 1]])
        for frac in (0.03, 0.5, 0.77):
            k = int(frac * len(mem))
            i = int(mem[k])
            if i >= N:
                continue  # a static member: `adv` covers those
            m = glob[np.delete(mem, k), :3].astype(np.float64).mean(0)
            R, t = T[base.tformIdPerPoint[i]][:, :3], T[base.tformIdPerPoint[i]][:, 3]
            adv_mid.localPoints[i, :3] = (R.T @ (m - t)).astype(np.float32)
            moved += 1
    assert moved >= 6

    for prob, expect_fallbacks in ((base, False), (adv, True), (adv_mid, True)):
        g_p = g if prob is not adv_mid else orc.transform_points(table, prob.localPoints, prob.tformIdPerPoint)
        glob_p = np.concatenate([g_p, prob.staticPoints]).astype(np.float32)
        ref = orc.Gaussians(glob_p, np.concatenate([prob.ringIds, prob.staticRingIds]), prob.minGridSize, s)
        pbase = prob.getPoseParameters()
        params = np.stack([pbase] + [pbase + H_INCR * np.eye(len(pbase))[k] for k in range(8)])  # 9 evaluations: two sub-batches
        e_ref = None
        # long_split = 4096: every Gaussian of the latency tier hands its second pass to helper workgroups (serial_kernels.hip) -- the last helper
        # tests the bounds and, where they fail (`adv`), runs the chain itself
        # serial_tree = 3: the latency tier's block-wise fall-back with EVERY block of every long Gaussian chained member by member from the running
        # sum of the blocks before it (the live path, mode 1, chains only the blocks whose bounds fail -- on `adv` the ones with the planted members)
        for mode, split in (("1", 0), ("2", 0), ("0", 0), ("3", 0), ("1", 4096), ("2", 4096), ("3", 4096)):
            opt = hip.DmsaOptimizer(debug={"serial_tree": int(mode), "long_split": split})
            opt.upload(prob)
            opt.poseTables(pbase[None, :], download=False)
            opt.updateGlobalPoints(0, download=False)
            assert opt.buildGaussians(s) == (ref.M, ref.Mm)
            tables = opt.poseTables(params)
            opt.serialFallbackSums(reset=True)
            e = opt.evalResiduals(len(params))
            fallbacks = opt.serialFallbackSums()
            if e_ref is None:
                e_ref = []
                for b in range(len(params)):
                    gb = orc.transform_points(tables[b], prob.localPoints, prob.tformIdPerPoint)
                    e_ref.append(ref.residuals(np.concatenate([gb, prob.staticPoints]).astype(np.float32)))
                e_ref = np.array(e_ref)
            assert np.array_equal(e, e_ref), (mode, split, np.abs(e - e_ref).max())
            if mode == "1":
                chained = int((np.diff(ref.seg_offset) > 128).sum())  # Gaussians of the chain tiers, one or two sub-batches each
                assert (fallbacks > 0) == expect_fallbacks and fallbacks < 0.02 * chained, (fallbacks, chained)
            else:
                assert fallbacks == 0  # only the live test counts
            opt.close()


def test_window_beyond_two_million_points(hip, orc):
    """2.7 M points: more 1024-point blocks than the lattice kernel keeps in LDS (2048; the rest of the block bounds are read from
    global memory), 330 sort tiles per level.  Voxel structure and Gaussian sets bit-exact against the oracle."""
    prob = synth.window_problem(seed=5, scans=10, rings=128, az_steps=2048, num_static=100_000)
    s = DmsaOptimSettings.sliding_window()
    n = prob.localPoints.shape[0] + prob.staticPoints.shape[0]
    assert n > 2048 * 1024
    table, _ = orc.window_pose_table(prob)
    g = orc.transform_points(table, prob.localPoints, prob.tformIdPerPoint)
    glob = np.concatenate([g, prob.staticPoints]).astype(np.float32)
    ref = orc.Gaussians(glob, np.concatenate([prob.ringIds, prob.staticRingIds]), prob.minGridSize, s)
    opt = hip.DmsaOptimizer()
    opt.upload(prob)
    opt.poseTables(prob.getPoseParameters(), download=False)
    opt.updateGlobalPoints(0, download=False)
    assert opt.buildGaussians(s) == (ref.M, ref.Mm)
    for level, f in ((0, s.grid_size_1_factor), (1, s.grid_size_2_factor)):
        info, code, key, order = opt.voxelLevel(level)
        info_r, code_r, key_r, order_r = orc.voxelize(glob, float(np.float32(f) * np.float32(prob.minGridSize)))
        assert (info.depth, info.num_events, info.num_leaves) == (info_r.depth, info_r.num_events, info_r.num_leaves)
        assert np.array_equal(code, code_r) and np.array_equal(order, order_r)
    seg, memb, info12, w = opt.gaussians()
    assert np.array_equal(seg, ref.seg_offset) and np.array_equal(memb, ref.members)
    assert np.array_equal(info12, ref.info) and np.array_equal(w, ref.weights)
    opt.close()


@pytest.mark.parametrize("compress", ["1", "0"])
def test_fine_grid_and_full_width_codes(hip, orc, compress):
    """minGridSize = 0.012 m: trees of depth >= 11.  With key_compress = 0 (include/dmsa_debug.h) the leaf codes keep all 3 x depth bits and no longer fit 32
    bits -- the 64-bit code path (library sort, 64-bit leaf segmentation), which is also what a mis-predicted code range falls back to.
    Same structure, bit for bit, either way."""
    debug = {"key_compress": int(compress)}
    prob = synth.window_problem(seed=21, scans=3, rings=64, az_steps=512, num_static=20_000)
    prob.minGridSize = 0.012
    s = DmsaOptimSettings.sliding_window(num_iter=2)
    table, _ = orc.window_pose_table(prob)
    g = orc.transform_points(table, prob.localPoints, prob.tformIdPerPoint)
    glob = np.concatenate([g, prob.staticPoints]).astype(np.float32)
    ref = orc.Gaussians(glob, np.concatenate([prob.ringIds, prob.staticRingIds]), prob.minGridSize, s)
    opt = hip.DmsaOptimizer(debug=debug)
    opt.upload(prob)
    opt.poseTables(prob.getPoseParameters(), download=False)
    opt.updateGlobalPoints(0, download=False)
    assert opt.buildGaussians(s) == (ref.M, ref.Mm)
    info0 = opt.voxelLevel(0)[0]
    assert info0.depth >= 11  # 3 x 11 bits + the marker of non-finite points + the level tag: beyond 32 bits when uncompressed
    for level, f in ((0, s.grid_size_1_factor), (1, s.grid_size_2_factor)):
        info, code, key, order = opt.voxelLevel(level)
        info_r, code_r, key_r, order_r = orc.voxelize(glob, float(np.float32(f) * np.float32(prob.minGridSize)))
        assert info.depth == info_r.depth and info.num_leaves == info_r.num_leaves
        assert np.array_equal(order, order_r)
    seg, memb, info12, w = opt.gaussians()
    assert np.array_equal(seg, ref.seg_offset) and np.array_equal(memb, ref.members) and np.array_equal(info12, ref.info)
    opt.close()
    _parity_run(hip, orc, prob, s, debug=debug)


def test_separate_level_sorts_on_an_odd_point_count(hip, orc):
    """merge_sort = 0 (include/dmsa_debug.h) sorts the two resolutions separately, as large windows do; with an odd point count the level-1 arrays start at an
    address that is only 4-byte aligned (views into the shared key / index arrays)."""
    prob = synth.window_problem(seed=23, scans=3, rings=32, az_steps=256, num_static=5001)
    assert (prob.localPoints.shape[0] + prob.staticPoints.shape[0]) % 2 == 1
    _parity_run(hip, orc, prob, DmsaOptimSettings.sliding_window(num_iter=3), debug={"merge_sort": 0})
    _parity_run(hip, orc, prob, DmsaOptimSettings.sliding_window(num_iter=2), debug={"merge_sort": 0, "fused_segments": 0, "dual_stream": 0, "fused_leaf_scan": 0})


def test_pow_minus_one_of_the_counts_is_the_host_libms_powf(hip):
    """Gaussians.h:172: Eigen evaluates pow(-1) of the member counts with libm's powf per coefficient, which is one ulp away from the
    correctly rounded 1.0f / n for some counts (953, 2071, ... with glibc >= 2.27).  The device divides and applies the differences the
    host's libm reports: every count up to 2^21, bit for bit."""
    import ctypes as C
    libm = C.CDLL("libm.so.6")
    libm.powf.argtypes, libm.powf.restype = [C.c_float, C.c_float], C.c_float
    n = np.arange(1, 1 << 21, dtype=np.int32)
    want = np.array([libm.powf(float(v), -1.0) for v in n[:200_000]], np.float32)
    got = hip.DmsaOptimizer().powMinusOne(n)
    assert np.array_equal(got[:200_000].view(np.int32), want.view(np.int32))
    div = (np.float32(1.0) / n.astype(np.float32)).astype(np.float32)
    differing = n[got.view(np.int32) != div.view(np.int32)]
    for v in differing[:50].tolist() + differing[-50:].tolist():  # the counts where the table matters, also beyond the first 200 000
        assert np.float32(libm.powf(float(v), -1.0)).view(np.int32) == got[v - 1].view(np.int32), v
    print(f"[powf] {differing.size} of {n.size} counts differ from the division on this libm, first {differing[:6].tolist()}")


def test_fit_follows_the_l1_size_of_the_reference_machine(hip, orc):
    """Eigen sizes the depth blocks of centered^T * centered from the L1d of the machine it runs on: with eigen_l1_bytes = 48 KB (blocks of 1016
    members instead of 680) the device's information matrices are the oracle's for that size -- and differ from the 32 KB ones."""
    prob = synth.rosette_window_problem(seed=2, scans=4, pts_per_scan=6000, num_static=3000)
    s = DmsaOptimSettings.sliding_window()
    table, _ = orc.window_pose_table(prob)
    g = orc.transform_points(table, prob.localPoints, prob.tformIdPerPoint)
    glob = np.concatenate([g, prob.staticPoints]).astype(np.float32)
    ids = np.concatenate([prob.ringIds, prob.staticRingIds])
    ref32 = orc.Gaussians(glob, ids, prob.minGridSize, s)
    assert np.diff(ref32.seg_offset).max() > 680
    try:
        orc.set_eigen_l1_bytes(48 * 1024)
        ref48 = orc.Gaussians(glob, ids, prob.minGridSize, s)
    finally:
        orc.set_eigen_l1_bytes(32 * 1024)
    assert not np.array_equal(ref32.info, ref48.info)
    for l1, ref in ((32 * 1024, ref32), (48 * 1024, ref48)):
        opt = hip.DmsaOptimizer(debug={"eigen_l1_bytes": l1})
        opt.upload(prob)
        opt.poseTables(prob.getPoseParameters(), download=False)
        opt.updateGlobalPoints(0, download=False)
        assert opt.buildGaussians(s) == (ref.M, ref.Mm)
        _, _, info, w = opt.gaussians()
        assert np.array_equal(info, ref.info) and np.array_equal(w, ref.weights), l1
        opt.close()
