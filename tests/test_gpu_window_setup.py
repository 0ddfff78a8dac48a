"""GPU side of SURVEY.md 8(f) row f3: the per-point tformIdPerPoint search (bit-exact against std::lower_bound) and whole
sequences — prepareTrajectoryForOptimization -> optimizeSet -> next window's initial guess — against the oracle's chain."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import posemath, synth
from dmsa_lidar_slam_amd import window_setup as ws
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

pytestmark = pytest.mark.gpu

GYR_COV, ACC_COV = np.diag([1e-4] * 3), np.diag([1e-2] * 3)


@pytest.fixture(scope="module")
def setup():
    s = ws.WindowSetup(device=0)
    yield s
    s.close()


@pytest.mark.parametrize("dt_res,n", [(1e-3, 1_310_720), (1e-4, 200_000), (1e-3, 1), (2e-3, 77)])
def test_tform_indices(setup, orc, dt_res, n):
    """registerPcBuffer :240-260 at the bench size (10 x 131 072 stamps, n_total = 1002: grid in LDS) and with a 10 002-entry grid
    (global-memory path); stamps outside the window, exactly on grid values, NaN and infinities included."""
    rng = np.random.default_rng(n)
    ow = orc.WindowSetup()
    s = setup.initTraj(1.6e9 + 0.25, 1.6e9 + 1.25, 6, False, dt_res)
    st = 1.6e9 + 0.25 + rng.uniform(-0.01, 1.02, n)
    k = min(n, s.n_total)
    st[:k] = s.t0 + s.trajTime[:k]  # exactly representable hits are rare (t0 + grid rounds), near-hits are the point
    if n > 10:
        st[-3:] = [np.nan, np.inf, -np.inf]
    got = setup.tformIdPerPoint(s, st)
    assert np.array_equal(got, ow.tformIdPerPoint(s, st))
    assert got.min() >= 0 and got.max() <= s.n_total - 1
    assert setup.tformIdPerPoint(s, np.zeros(0)).shape == (0,)


def _imu_buffer(cls, traj, rng_seed):
    st, acc, ang = synth.imu_stream(traj, -0.3, 1.2, rate=400.0, rng=np.random.default_rng(rng_seed), sigma_acc=0.02, sigma_gyr=0.002)
    buf = cls(10000)
    for t in st[0] - (50 - np.arange(50)) * 0.0025:  # at rest while the gyro bias is estimated (ImuBuffer.h:60-64)
        buf.addMeasurement([0.0, 0.0, 9.805], np.zeros(3), t)
    for t, a, w in zip(st, acc, ang):
        buf.addMeasurement(a, w, t)
    return buf


@pytest.mark.parametrize("use_imu", [False, True])
def test_sequence_of_windows_matches_oracle_chain(setup, orc, use_imu):
    """Three consecutive 5-scan windows.  Each window: initTraj -> (IMU transfer + preintegration) -> updateInitialGuess from the
    previous window's optimised poses -> registerPcBuffer -> optimizeSet (parity path).  The oracle runs the same chain with its own
    setup and optimiser: the problems handed to optimizeSet must be bit-identical, the optimised poses within 1e-4 m / 1e-4 rad."""
    clouds, traj = synth.scan_sequence(seed=2, scans=7, rings=32, az_steps=256)
    settings = DmsaOptimSettings.sliding_window(use_imu=use_imu, num_iter=3)
    gpu = DmsaOptimizer(device=0)
    ow = orc.WindowSetup()
    buf_p, buf_o = (_imu_buffer(ws.ImuBuffer, traj, 9), _imu_buffer(orc.ImuBuffer, traj, 9)) if use_imu else (None, None)
    old_p = old_o = None
    init_p = init_o = False
    for w in range(3):
        win = clouds[w: w + 5]
        tr_p, prob_p, init_p = setup.prepareTrajectoryForOptimization(win, old_p, init_p, 6, 1e-3, buf_p, GYR_COV, ACC_COV)
        # the oracle's chain
        t_min, t_max = min(c[1].min() for c in win), max(c[1].max() for c in win)
        tr_o = ow.initTraj(t_min, t_max, 6, use_imu, 1e-3)
        if use_imu:
            ow.transferImuMeasurements(tr_o, buf_o)
            ow.updatePreintFactors(tr_o, GYR_COV, ACC_COV)
        init_o = ow.updateInitialGuess(init_o, tr_o, old_o, use_imu)
        prob_o = ws.assemble_problem(tr_o, win, ow.tformIdPerPoint(tr_o, np.concatenate([c[1] for c in win])))
        assert init_p and init_o
        for f in ("relOrientations", "relTranslations", "stamps", "trajTime", "tformIdPerPoint", "localPoints", "ringIds"):
            assert np.array_equal(getattr(prob_p, f), getattr(prob_o, f)), (w, f)  # same old window in, same problem out: bit for bit
        if use_imu:
            for f in ("paramIndices", "preintImuRots", "preintRelPositions", "preintRelVelocity", "CovPVRot_inv"):
                assert np.array_equal(getattr(prob_p, f), getattr(prob_o, f)), (w, f)
        if w == 0:  # the very first window starts from rest in the reference; give both chains a converged first window instead
            R, p = traj.pose(tr_p.t0 - 1.6e9 + tr_p.stamps)
            ro, rt = posemath.global2relative(R.as_rotvec(), p)
            for prob in (prob_p, prob_o):
                prob.relOrientations[...], prob.relTranslations[...] = ro, rt
        rep = gpu.optimizeSet(prob_p, settings)
        rep_o, _, _ = orc.optimize_window(prob_o, settings)
        assert rep.iterations == rep_o.iterations and rep.num_gaussians == rep_o.num_gaussians
        assert np.abs(prob_p.relTranslations - prob_o.relTranslations).max() < 1e-4 and np.abs(prob_p.relOrientations - prob_o.relOrientations).max() < 1e-4
        # the optimised window becomes oldTraj.  Both chains continue from the GPU's poses: the optimiser is ill-conditioned enough
        #  that 1e-9 (1e-6 with IMU rows) between the two results would grow by ~100x per window and blur the comparison
        for tr in (tr_p, tr_o):
            tr.relOrientations[...], tr.relTranslations[...] = prob_p.relOrientations, prob_p.relTranslations
        old_p, old_o = tr_p, tr_o
        if w > 0:  # sanity: the chain follows the 1 m/s motion (no static map, 32 rings, 3 iterations: loose bound)
            _, p = traj.pose(tr_p.t0 - 1.6e9 + tr_p.stamps)
            go, gt = posemath.relative2global(prob_p.relOrientations, prob_p.relTranslations)
            assert np.abs(gt - p).max() < 0.3
    gpu.close()
