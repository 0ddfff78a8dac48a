"""The committed fixture of SURVEY 8(f) rows f1-f4 (tests/golden/next_rows_small.npz, written by tests/golden/make_golden.py): the
oracle must reproduce it (CPU), and so must the HIP path through the C ABI (GPU) — without a live comparison against the oracle."""
import os

import numpy as np
import pytest

from dmsa_lidar_slam_amd import window_setup as ws
from dmsa_lidar_slam_amd.static_points import StaticSelectProblem
from dmsa_lidar_slam_amd.wire_formats import PointCloud2Msg

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "next_rows_small.npz"))
f32 = np.float32


def _select_problem():
    return StaticSelectProblem(windowPoints=Z["sp_windowPoints"], keyframeIds=Z["sp_keyframeIds"], frameOffsets=Z["sp_frameOffsets"], keyPoints=Z["sp_keyPoints"],
                               keyNormals=Z["sp_keyNormals"], keyRingIds=Z["sp_keyRingIds"], currPos=Z["sp_currPos"], minGridSize=float(Z["sp_minGridSize"]))


def _check_static(sel, pick, overlap):
    assert np.array_equal(sel.staticPoints, Z["sp_staticPoints"]) and np.array_equal(sel.staticIds, Z["sp_staticIds"])
    assert np.array_equal(sel.overlapPerKeyframe, Z["sp_overlapPerKeyframe"])
    assert [sel.keyframeId, sel.minRelatedKeyId, sel.maxOverlap] == Z["sp_ids"].tolist()
    assert np.array_equal(pick, Z["sp_thin_pick"]) and list(overlap) == Z["sp_overlap"].tolist()
    assert sel.staticPoints.shape[0] > 200


def _check_preprocess(out):
    assert np.array_equal(out[0], Z["pre_xyz"]) and np.array_equal(out[1], Z["pre_src"]) and f32(out[2]) == Z["pre_grid"] and out[0].shape[0] > 300


def _msg():
    h, w, step = (int(v) for v in Z["pc2_meta"])
    return PointCloud2Msg(height=h, width=w, point_step=step, field_offsets=Z["pc2_offsets"], data=Z["pc2_data"], stamp=float(Z["pc2_stamp"]))


def _window_setup(setup, buffer_cls):
    buf = buffer_cls(10000)
    for t, a, w in zip(Z["ws_imu_stamps"], Z["ws_imu_acc"], Z["ws_imu_ang"]):
        buf.addMeasurement(a, w, t)
    cur = setup.initTraj(float(Z["ws_t_min"]), float(Z["ws_t_max"]), 6, True, 1e-3)
    setup.transferImuMeasurements(cur, buf)
    setup.updatePreintFactors(cur, Z["ws_gyr_cov"], Z["ws_acc_cov"])
    old = setup.initTraj(float(Z["ws_old_t0"]), float(Z["ws_old_t_max"]), 6, True, 1e-3)
    old.relOrientations[...], old.relTranslations[...] = Z["ws_old_rel_o"], Z["ws_old_rel_t"]
    assert setup.updateInitialGuess(True, cur, old, True)
    for name, got in (("trajTime", cur.trajTime), ("stamps", cur.stamps), ("paramIndices", cur.paramIndices), ("accMeas", cur.accMeas),
                      ("angVelMeas", cur.angVelMeas), ("preintImuRots", cur.preintImuRots), ("preintRelPositions", cur.preintRelPositions),
                      ("preintRelVelocity", cur.preintRelVelocity), ("CovPVRot_inv", cur.CovPVRot_inv), ("preintPosComplHor", cur.preintPosComplHor),
                      ("guess_rel_o", cur.relOrientations), ("guess_rel_t", cur.relTranslations)):
        assert np.array_equal(got, Z["ws_" + name]), name
    return cur


def test_oracle_reproduces_next_rows_golden(orc):
    sp = _select_problem()
    sel = orc.select_static_points(sp)
    pick = orc.random_grid_downsampling(sel.staticPoints, f32(sp.minGridSize) / f32(2.0), 7)
    _check_static(sel, pick, orc.get_overlap(sel.staticPoints[pick], sp.windowPoints, sp.minGridSize))
    _check_preprocess(orc.preprocess_scan(Z["pre_raw"], 9, 1000, 3.0, 0.5, Z["pre_T"]))
    ow = orc.WindowSetup()
    cur = _window_setup(ow, orc.ImuBuffer)
    assert np.array_equal(ow.tformIdPerPoint(cur, Z["ws_point_stamps"]), Z["ws_tform_idx"])
    x, s, i = orc.decode_pointcloud2(_msg(), "velodyne")
    assert np.array_equal(x, Z["pc2_xyz"]) and np.array_equal(s, Z["pc2_stamps"]) and np.array_equal(i, Z["pc2_ids"])
    assert [orc.format_tum_pose(q[0], q[1:4], q[4:7]) for q in Z["tum_poses"]] == Z["tum_lines"].tolist()
    nrm, nn = orc.update_normals(Z["kf_xyz"], neighbours=True)
    assert np.array_equal(nn, Z["kf_neighbours"]) and np.array_equal(nrm, Z["kf_normals"], equal_nan=True)


def test_product_host_functions_reproduce_next_rows_golden():
    """The host half of the product (window setup, TUM lines) needs no GPU."""
    from dmsa_lidar_slam_amd import wire_formats as wf

    _window_setup(ws.WindowSetup(0), ws.ImuBuffer)
    assert [wf.addPoseToFile(q[0], q[1:4], q[4:7]) for q in Z["tum_poses"]] == Z["tum_lines"].tolist()


@pytest.mark.gpu
def test_hip_reproduces_next_rows_golden():
    from dmsa_lidar_slam_amd import wire_formats as wf
    from dmsa_lidar_slam_amd.keyframe_cloud import KeyframeCloudBuilder
    from dmsa_lidar_slam_amd.static_points import StaticPointSelector

    g = StaticPointSelector(0)
    sp = _select_problem()
    sel = g.selectStaticPoints(sp)
    pick = g.randomGridDownsampling(sel.staticPoints, f32(sp.minGridSize) / f32(2.0), 7)
    _check_static(sel, pick, g.getOverlap(sel.staticPoints[pick], sp.windowPoints, sp.minGridSize))
    _check_preprocess(g.preProcess(Z["pre_raw"], 9, 1000, 3.0, 0.5, Z["pre_T"]))
    g.close()
    setup = ws.WindowSetup(0)
    cur = _window_setup(setup, ws.ImuBuffer)
    assert np.array_equal(setup.tformIdPerPoint(cur, Z["ws_point_stamps"]), Z["ws_tform_idx"])
    setup.close()
    dec = wf.PointCloud2Decoder("velodyne")
    x, s, i = dec.decode(_msg())
    assert np.array_equal(x, Z["pc2_xyz"]) and np.array_equal(s, Z["pc2_stamps"]) and np.array_equal(i, Z["pc2_ids"])
    dec.close()
    kb = KeyframeCloudBuilder(0)
    nrm, nn = kb.updateNormals(Z["kf_xyz"], 0.3, neighbours=True)
    kb.close()
    assert np.array_equal(nn, Z["kf_neighbours"]) and np.array_equal(nrm, Z["kf_normals"], equal_nan=True)
