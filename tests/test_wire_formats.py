"""SURVEY.md 8(f) row f4 on the CPU: the oracle's PointCloud2 decoding against a numpy structured-dtype encoder, the TUM line and
the non-keyframe pose composition against scipy, and the product's host functions against the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from dmsa_lidar_slam_amd import wire_formats as wf
from wire_util import LAYOUTS, expected_unknown, make_msg


@pytest.mark.parametrize("sensor", [s for s in LAYOUTS if s != "unknown"])
def test_oracle_decoder_reads_every_sensor_layout(orc, sensor):
    msg, exp = make_msg(sensor, 5000, seed=3, height=5)
    xyz, st, ids = orc.decode_pointcloud2(msg, sensor)
    assert np.array_equal(xyz[:, 0], exp["x"]) and np.array_equal(xyz[:, 1], exp["y"]) and np.array_equal(xyz[:, 2], exp["z"]) and not xyz[:, 3].any()
    assert np.array_equal(st, exp["stamp"]) and np.array_equal(ids, exp["id"])


def test_oracle_decoder_unknown_sensor_heuristic(orc):
    msg, exp = make_msg("unknown", 2500, seed=4)
    xyz, st, ids = orc.decode_pointcloud2(msg, "unknown", delta_t_pcs=0.1003)
    es, ei = expected_unknown(2500, msg.stamp, 0.1003)
    assert np.array_equal(st, es) and np.array_equal(ids, ei) and np.array_equal(xyz[:, 0], exp["x"])


@pytest.mark.parametrize("seed", range(6))
def test_tum_line(orc, seed):
    rng = np.random.default_rng(seed)
    stamp = 1.6e9 + rng.uniform(0, 1e4)
    pos = rng.normal(0, 50, 3)
    orient = rng.normal(0, 1.0, 3) if seed else np.array([0.0, 0.0, np.pi - 1e-9])  # seed 0: the largest-diagonal branch of Quaterniond(R)
    line = orc.format_tum_pose(stamp, pos, orient)
    assert line == wf.addPoseToFile(stamp, pos, orient)
    tok = line.split()
    assert line.endswith("\n") and len(tok) == 8
    assert tok[0] == f"{stamp:.6f}" and tok[1:4] == [f"{v:.5f}" for v in pos]
    q = np.array([float(v) for v in tok[4:]])  # x y z w
    qs = Rot.from_rotvec(orient).as_quat()
    assert np.allclose(q, qs, atol=1e-6) or np.allclose(q, -qs, atol=1e-6)
    assert all(len(t.split(".")[1]) == d for t, d in zip(tok, (6, 5, 5, 5, 6, 6, 6, 6)))


def test_nonkeyframe_pose_composition(orc):
    rng = np.random.default_rng(1)
    for _ in range(20):
        kp, ko, t, o = rng.normal(0, 10, 3), rng.normal(0, 0.8, 3), rng.normal(0, 2, 3), rng.normal(0, 0.5, 3)
        gp, go = orc.compose_nonkeyframe_pose(kp, ko, t, o)
        pp, po = wf.composeNonKeyframePose(kp, ko, t, o)
        assert np.array_equal(gp, pp) and np.array_equal(go, po)
        R = Rot.from_rotvec(ko)
        assert np.allclose(gp, R.apply(t) + kp, atol=1e-12)
        assert np.allclose(Rot.from_rotvec(go).as_matrix(), (R * Rot.from_rotvec(o)).as_matrix(), atol=1e-12)
