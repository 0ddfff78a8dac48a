"""include/dmsa_window_ring.h: the scans of the sliding window resident in HBM (RingBuffer.h:31-88).  The window assembled on the device
from the ring -- scans oldest first, pose-table rows from the resident stamps (registerPcBuffer, ContinuousTrajectory.h:240-260) -- must be
the problem dmsa_window_upload builds from host arrays, bit for bit: same optimised poses, same trace.  Then the window slides: one scan is
pushed, the oldest leaves, and the result equals a from-scratch upload of the new window."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

pytestmark = pytest.mark.gpu


def _subwindow(full, first_scan, num_scans, dt_res=1e-3, num_control_poses=6):
    """The window of scans [first_scan, first_scan + num_scans) of a longer recording: its own time grid (initTraj), rows (registerPcBuffer)
    and control poses (the long problem's truth, perturbed deterministically)."""
    from dmsa_lidar_slam_amd.posemath import global2relative
    from dmsa_lidar_slam_amd.problems import ContinuousTrajectory

    a, b = int(full.scanOffsets[first_scan]), int(full.scanOffsets[first_scan + num_scans])
    stamps = full.pointStamps[a:b]
    t0 = stamps.min()
    horizon = stamps.max() - t0 + dt_res
    n_total = int(round(horizon / dt_res)) + 1
    traj_time = np.linspace(0.0, horizon, n_total)
    ctrl = np.linspace(0.0, horizon, num_control_poses)
    rows = np.minimum(np.searchsorted(traj_time, stamps - t0, side="left"), n_total - 1).astype(np.int32)
    traj = synth.SmoothTrajectory(p0=np.array([4.0, 3.0, 1.5]))
    R, p = traj.pose((t0 - 1.6e9) + ctrl)
    rng = np.random.default_rng(100 + first_scan)
    go, gt = R.as_rotvec(), p.copy()
    go[1:] += rng.normal(0, 0.005, go[1:].shape)
    gt[1:] += rng.normal(0, 0.02, gt[1:].shape)
    ro, rt = global2relative(go, gt)
    w = ContinuousTrajectory(relOrientations=ro, relTranslations=rt, stamps=ctrl, trajTime=traj_time, localPoints=full.localPoints[a:b], tformIdPerPoint=rows,
                             ringIds=full.ringIds[a:b], staticPoints=full.staticPoints, staticRingIds=full.staticRingIds, minGridSize=full.minGridSize)
    return w, float(t0)


def test_window_from_the_ring_equals_the_host_upload_and_slides(hip, orc):
    scans_total, scans_win = 6, 4
    full = synth.window_problem(seed=41, scans=scans_total, rings=32, az_steps=256, num_static=3000)
    s = DmsaOptimSettings.sliding_window(num_iter=3)
    per_scan = int(np.diff(full.scanOffsets).max())
    ring = hip.DmsaOptimizer()
    ring.ringCreate(scans_win, per_scan, full.staticPoints.shape[0], 2000)

    def push(k):
        a, b = int(full.scanOffsets[k]), int(full.scanOffsets[k + 1])
        ring.ringPush(full.localPoints[a:b], full.pointStamps[a:b], full.ringIds[a:b])

    for k in range(scans_win):
        push(k)
    for first in range(scans_total - scans_win + 1):
        if first > 0:
            push(first + scans_win - 1)  # the window slides by one scan: one upload
        assert ring.ringPoints() == (scans_win, int(full.scanOffsets[first + scans_win] - full.scanOffsets[first]))
        w, t0 = _subwindow(full, first, scans_win)
        # resident path
        wr = w.copy()
        ring.uploadFromRing(wr, t0)
        rep_r = ring.optimizeResident(s)
        ro_r, rt_r = ring.poses()
        tr_r = ring.trace()
        # host-array path on a fresh context, and the oracle
        wh = w.copy()
        host = hip.DmsaOptimizer()
        rep_h = host.optimizeSet(wh, s)
        tr_h = host.trace()
        host.close()
        assert (rep_r.iterations, rep_r.stop_reason, rep_r.num_gaussians, rep_r.num_memberships, rep_r.evaluations) == \
               (rep_h.iterations, rep_h.stop_reason, rep_h.num_gaussians, rep_h.num_memberships, rep_h.evaluations)
        assert [(t["M"], t["Mm"], t["best_k"], t["error0"], t["step_norm"]) for t in tr_r[: rep_r.iterations]] == \
               [(t["M"], t["Mm"], t["best_k"], t["error0"], t["step_norm"]) for t in tr_h[: rep_h.iterations]]
        assert np.array_equal(ro_r, wh.relOrientations) and np.array_equal(rt_r, wh.relTranslations)
        wo = w.copy()
        rep_o, _, _ = orc.optimize_window(wo, s)
        assert rep_o.iterations == rep_r.iterations
        assert np.array_equal(ro_r, wo.relOrientations) and np.array_equal(rt_r, wo.relTranslations)
    ring.close()


def test_ring_rejects_what_does_not_fit(hip):
    opt = hip.DmsaOptimizer()
    with pytest.raises(hip.DmsaError):
        opt.ringPush(np.zeros((4, 4), np.float32), np.zeros(4), np.zeros(4, np.int32))  # no ring yet
    opt.ringCreate(2, 100, 10, 500)
    with pytest.raises(hip.DmsaError):
        opt.ringPush(np.zeros((101, 4), np.float32), np.zeros(101), np.zeros(101, np.int32))  # larger than a slot
    opt.ringPush(np.zeros((0, 4), np.float32), np.zeros(0), np.zeros(0, np.int32))  # an empty scan is a scan
    assert opt.ringPoints() == (1, 0)
    opt.close()
