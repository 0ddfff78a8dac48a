"""include/dmsa_aos.h: the reference's own point containers at the ABI -- pcl::PointCloud<PointStampId> (32-byte points, one cloud per scan
of the ring buffer, PointStampId.h:33-45) and pcl::PointCloud<pcl::PointNormal> (48-byte points, one cloud per keyframe) handed over as
they lie in memory, packed on the device.  Same poses, same traces, same global points as the flat entry points, bit for bit."""
import numpy as np
import pytest

from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

pytestmark = pytest.mark.gpu


def _trace_key(t):
    return [(x["M"], x["M1"], x["Mm"], x["best_k"], x["error0"], x["step_norm"]) for x in t]


def test_window_from_strided_scan_clouds(hip):
    prob = synth.window_problem(seed=31, scans=4, rings=32, az_steps=256, num_static=4000)
    assert len(prob.scanOffsets) == 5
    s = DmsaOptimSettings.sliding_window(num_iter=4)
    flat, aos = prob.copy(), prob.copy()
    o1, o2 = hip.DmsaOptimizer(), hip.DmsaOptimizer()
    r1 = o1.optimizeSet(flat, s)
    r2 = o2.optimizeSetAos(aos, s, reserve=True)
    assert (r1.iterations, r1.stop_reason, r1.num_gaussians, r1.num_memberships, r1.error0) == (r2.iterations, r2.stop_reason, r2.num_gaussians, r2.num_memberships, r2.error0)
    assert _trace_key(o1.trace()) == _trace_key(o2.trace())
    assert np.array_equal(flat.relOrientations, aos.relOrientations) and np.array_equal(flat.relTranslations, aos.relTranslations)
    g1, g2 = o1.globalPoints(), o2.globalPointsAos()
    assert np.array_equal(g1[:, 0], g2["x"]) and np.array_equal(g1[:, 1], g2["y"]) and np.array_equal(g1[:, 2], g2["z"])
    assert np.all(g2["stamp"] == 7.0) and np.all(g2["id"] == 123)  # the call writes x, y, z and nothing else of a point
    # a second call on the same context (staging and device buffers reused), with a bad tform index: refused, not packed
    bad = prob.copy()
    bad.tformIdPerPoint = bad.tformIdPerPoint.copy()
    bad.tformIdPerPoint[5] = bad.trajTime.shape[0] + 3
    with pytest.raises(hip.DmsaError, match="tform_idx"):
        o2.optimizeSetAos(bad, s)
    again = prob.copy()
    o2.optimizeSetAos(again, s)
    assert np.array_equal(flat.relOrientations, again.relOrientations) and np.array_equal(flat.relTranslations, again.relTranslations)
    o1.close(), o2.close()


def test_keyframes_from_strided_point_normal_clouds(hip):
    prob = synth.keyframe_problem(seed=4, frames=13, rings=16, az_steps=96, arc=0.8)
    s = DmsaOptimSettings.keyframe_map(num_iter=3)
    flat, aos = prob.copy(), prob.copy()
    o1, o2 = hip.DmsaOptimizer(), hip.DmsaOptimizer()
    r1, r2 = o1.optimizeSet(flat, s), o2.optimizeSetAos(aos, s)
    assert (r1.iterations, r1.stop_reason, r1.num_gaussians, r1.num_memberships, r1.error0) == (r2.iterations, r2.stop_reason, r2.num_gaussians, r2.num_memberships, r2.error0)
    assert _trace_key(o1.trace()) == _trace_key(o2.trace())
    assert np.array_equal(flat.relOrientations, aos.relOrientations) and np.array_equal(flat.relTranslations, aos.relTranslations)
    g1, g2 = o1.globalPoints(), o2.globalPointsAos(keyframes=True)
    assert np.array_equal(g1[:, 0], g2["x"]) and np.array_equal(g1[:, 2], g2["z"])
    assert np.abs(g2["normal_x"] ** 2 + g2["normal_y"] ** 2 + g2["normal_z"] ** 2 - 1.0).max() < 1e-3  # rotated unit normals
    o1.close(), o2.close()


def test_resident_ring_fed_from_strided_scans(hip):
    """dmsa_window_ring_push_aos / dmsa_window_upload_from_ring_aos: the scans of the window stay in HBM, every scan arrives as the 32-byte
    points of its PCL cloud (coordinates, the double stamp at offset 16, the ring id at 24), the static points as the tail of globalPoints;
    the pose-table rows come from the resident stamps (registerPcBuffer on the device) -- the same problem, the same poses."""
    prob = synth.window_problem(seed=31, scans=4, rings=32, az_steps=256, num_static=4000)
    s = DmsaOptimSettings.sliding_window(num_iter=3)
    flat = prob.copy()
    o1 = hip.DmsaOptimizer()
    o1.optimizeSet(flat, s)
    off = prob.scanOffsets
    o2 = hip.DmsaOptimizer()
    o2.ringCreate(len(off) - 1, int(np.diff(off).max()), prob.staticPoints.shape[0], prob.trajTime.size, prob.relOrientations.shape[0])
    for k in range(len(off) - 1):
        o2.ringPushAos(prob.localPoints[off[k]:off[k + 1]], prob.pointStamps[off[k]:off[k + 1]], prob.ringIds[off[k]:off[k + 1]])
    ring = prob.copy()
    o2.uploadFromRingAos(ring, prob.t0)
    o2.optimizeResident(s)
    ro, rt = o2.poses()
    assert np.array_equal(flat.relOrientations, ro) and np.array_equal(flat.relTranslations, rt)
    o1.close(), o2.close()


def test_a_window_large_enough_for_sliced_transfers(hip):
    """Above 2^18 points the upload gathers and copies slice by slice and the write-back of the global points comes down in pieces
    (csrc/aos_upload.cpp): same poses and the same global points as the flat call, every piece in its place."""
    prob = synth.window_problem(seed=5, scans=3, rings=128, az_steps=1024, num_static=30_000)
    assert prob.localPoints.shape[0] + prob.staticPoints.shape[0] > (1 << 18) + 4096
    s = DmsaOptimSettings.sliding_window(num_iter=2)
    flat, aos = prob.copy(), prob.copy()
    o1, o2 = hip.DmsaOptimizer(), hip.DmsaOptimizer()
    o1.optimizeSet(flat, s)
    o2.optimizeSetAos(aos, s)
    assert np.array_equal(flat.relOrientations, aos.relOrientations) and np.array_equal(flat.relTranslations, aos.relTranslations)
    g1, g2 = o1.globalPoints(), o2.globalPointsAos()
    assert np.array_equal(g1[:, 0], g2["x"]) and np.array_equal(g1[:, 1], g2["y"]) and np.array_equal(g1[:, 2], g2["z"])
    assert np.all(g2["stamp"] == 7.0) and np.all(g2["id"] == 123)
    o1.close(), o2.close()
