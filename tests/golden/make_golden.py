"""Generates the golden fixtures in this directory.

The reference ships no tests or golden vectors for this path and cannot be built here, so these
fixtures are INPUT/OUTPUT DATA produced by the CPU oracle (oracle/dmsa_oracle.cpp) at the commit that introduced them:
they pin the oracle against regressions (tests/test_golden.py, CPU) and give the GPU parity tests expected outputs
that do not depend on a live oracle build.  Third-party pieces of the oracle are cross-checked independently against
scipy / an explicit PCL-octree model in tests/test_oracle_math.py and tests/test_oracle_voxel.py.

    python tests/golden/make_golden.py        # rewrites window_small.npz / keyframes_small.npz
    python tests/golden/make_golden.py next   # rewrites next_rows_small.npz only
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from dmsa_lidar_slam_amd import synth  # noqa: E402
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings  # noqa: E402
from oracle import oracle_py as orc  # noqa: E402


def window_case():
    prob = synth.window_problem(seed=21, scans=3, rings=16, az_steps=160, num_static=2500)
    s = DmsaOptimSettings.sliding_window(num_iter=4)
    table, dense = orc.window_pose_table(prob)
    g = orc.transform_points(table, prob.localPoints, prob.tformIdPerPoint)
    glob = np.concatenate([g, prob.staticPoints]).astype(np.float32)
    ids = np.concatenate([prob.ringIds, prob.staticRingIds])
    out = dict(relOrientations=prob.relOrientations, relTranslations=prob.relTranslations, stamps=prob.stamps, trajTime=prob.trajTime,
               localPoints=prob.localPoints, tformIdPerPoint=prob.tformIdPerPoint, ringIds=prob.ringIds, staticPoints=prob.staticPoints,
               staticRingIds=prob.staticRingIds, minGridSize=np.float32(prob.minGridSize), pose_table=table)
    for lvl, f in enumerate((s.grid_size_1_factor, s.grid_size_2_factor)):
        res = float(np.float32(f) * np.float32(prob.minGridSize))
        info, code, key, order = orc.voxelize(glob, res)
        out[f"vox{lvl}_code"], out[f"vox{lvl}_order"] = code, order
        out[f"vox{lvl}_meta"] = np.array([info.resolution, *info.min_xyz, info.depth, info.num_events, info.num_leaves, info.num_valid])
    G = orc.Gaussians(glob, ids, prob.minGridSize, s)
    out.update(seg_offset=G.seg_offset, members=G.members, info_mats=G.info, weights=G.weights, residuals0=G.residuals(glob))
    p = prob.copy()
    rep, gl, trace = orc.optimize_window(p, s, want_global=True)
    out.update(final_relOrientations=p.relOrientations, final_relTranslations=p.relTranslations,
               trace=np.array([[t["M"], t["M1"], t["Mm"], t["error0"], t["step_norm"], t["best_k"]] for t in trace]),
               report=np.array([rep.iterations, rep.stop_reason, rep.evaluations]))
    np.savez_compressed(os.path.join(HERE, "window_small.npz"), **out)


def keyframe_case():
    prob = synth.keyframe_problem(seed=5, frames=6, rings=16, az_steps=128, arc=0.4)
    s = DmsaOptimSettings.keyframe_map(num_iter=3)
    out = dict(relOrientations=prob.relOrientations, relTranslations=prob.relTranslations, frameOffsets=prob.frameOffsets,
               localPoints=prob.localPoints, localNormals=prob.localNormals, ringIds=prob.ringIds, minGridSize=np.float32(prob.minGridSize),
               measuredGravity=prob.measuredGravity, gravityPlausible=prob.gravityPlausible, pose_table=orc.keyframe_pose_table(prob),
               gravity_rows=orc.keyframe_additional_errors(prob))
    p = prob.copy()
    rep, gl, trace = orc.optimize_keyframes(p, s, want_global=True)
    out.update(final_relOrientations=p.relOrientations, final_relTranslations=p.relTranslations,
               trace=np.array([[t["M"], t["M1"], t["Mm"], t["error0"], t["step_norm"], t["best_k"]] for t in trace]),
               report=np.array([rep.iterations, rep.stop_reason, rep.evaluations]))
    np.savez_compressed(os.path.join(HERE, "keyframes_small.npz"), **out)


def next_rows_case():
    """inputs + the oracle's outputs of addStaticPoints, preProcess, the window setup, PointCloud2 decoding, TUM
    lines and keyframe normals on one small synthetic scene."""
    sys.path.insert(0, os.path.dirname(HERE))
    from wire_util import make_msg

    from dmsa_lidar_slam_amd import window_setup as ws

    rng = np.random.default_rng(31)
    out = {}
    # f1: static-point selection on a reduced window
    sp = synth.static_select_problem(seed=4, scans=2, rings=16, az_steps=128, frames=2, key_rings=16, key_az=96)
    sel = orc.select_static_points(sp)
    pick = orc.random_grid_downsampling(sel.staticPoints, np.float32(sp.minGridSize) / np.float32(2.0), 7)
    out.update(sp_windowPoints=sp.windowPoints, sp_keyframeIds=sp.keyframeIds, sp_frameOffsets=sp.frameOffsets, sp_keyPoints=sp.keyPoints,
               sp_keyNormals=sp.keyNormals, sp_keyRingIds=sp.keyRingIds, sp_currPos=sp.currPos, sp_minGridSize=np.float32(sp.minGridSize),
               sp_staticPoints=sel.staticPoints, sp_staticIds=sel.staticIds, sp_overlapPerKeyframe=sel.overlapPerKeyframe,
               sp_ids=np.array([sel.keyframeId, sel.minRelatedKeyId, sel.maxOverlap]), sp_thin_pick=pick,
               sp_overlap=np.array(orc.get_overlap(sel.staticPoints[pick], sp.windowPoints, sp.minGridSize)))
    # f2: preProcess of one raw scan
    clouds, traj = synth.scan_sequence(seed=8, scans=6, rings=16, az_steps=256)
    raw = np.concatenate([clouds[0][0], rng.uniform(0, 1, (clouds[0][0].shape[0], 1)).astype(np.float32)], axis=1)
    T = np.eye(4, dtype=np.float32)
    T[:3, 3] = [0.05, -0.12, 0.3]
    T[:3, :3] = np.array([[0.36, -0.93, 0.02], [0.93, 0.36, 0.01], [-0.02, 0.01, 1.0]], np.float32)
    xyz, src, grid = orc.preprocess_scan(raw, 9, 1000, 3.0, 0.5, T)
    out.update(pre_raw=raw, pre_T=T, pre_xyz=xyz, pre_src=src, pre_grid=np.float32(grid))
    # f3: window setup with IMU
    st, acc, ang = synth.imu_stream(traj, -0.3, 0.8, rate=400.0, rng=np.random.default_rng(32), sigma_acc=0.02, sigma_gyr=0.002)
    buf = orc.ImuBuffer(10000)
    for t, a, w in zip(st, acc, ang):
        buf.addMeasurement(a, w, t)
    win = clouds[:5]
    ow = orc.WindowSetup()
    cur = ow.initTraj(min(c[1].min() for c in win), max(c[1].max() for c in win), 6, True, 1e-3)
    ow.transferImuMeasurements(cur, buf)
    gyr, accc = np.diag([1e-4, 2e-4, 1.5e-4]), np.diag([1e-2, 2e-2, 1.5e-2])
    ow.updatePreintFactors(cur, gyr, accc)
    old = ow.initTraj(cur.t0 - 0.2, cur.t0 + 0.25, 6, True, 1e-3)
    R, p = traj.pose(old.t0 - 1.6e9 + old.stamps)
    from dmsa_lidar_slam_amd import posemath

    ro, rt = posemath.global2relative(R.as_rotvec(), p)
    old.relOrientations[...], old.relTranslations[...] = ro, rt
    ow.updateInitialGuess(True, cur, old, True)
    stamps = np.concatenate([c[1] for c in win])
    out.update(ws_imu_stamps=st, ws_imu_acc=acc, ws_imu_ang=ang, ws_t_min=cur.t0, ws_t_max=max(c[1].max() for c in win), ws_gyr_cov=gyr, ws_acc_cov=accc,
               ws_old_t0=old.t0, ws_old_t_max=cur.t0 + 0.25, ws_old_rel_o=ro, ws_old_rel_t=rt, ws_point_stamps=stamps,
               ws_trajTime=cur.trajTime, ws_stamps=cur.stamps, ws_paramIndices=cur.paramIndices, ws_accMeas=cur.accMeas, ws_angVelMeas=cur.angVelMeas,
               ws_preintImuRots=cur.preintImuRots, ws_preintRelPositions=cur.preintRelPositions, ws_preintRelVelocity=cur.preintRelVelocity,
               ws_CovPVRot_inv=cur.CovPVRot_inv, ws_preintPosComplHor=cur.preintPosComplHor, ws_guess_rel_o=cur.relOrientations,
               ws_guess_rel_t=cur.relTranslations, ws_tform_idx=ow.tformIdPerPoint(cur, stamps))
    # f4: wire formats + keyframe normals
    msg, _ = make_msg("velodyne", 2000, seed=5, height=4)
    dx, ds, di = orc.decode_pointcloud2(msg, "velodyne")
    out.update(pc2_data=msg.data, pc2_offsets=msg.field_offsets, pc2_meta=np.array([msg.height, msg.width, msg.point_step]), pc2_stamp=msg.stamp, pc2_xyz=dx,
               pc2_stamps=ds, pc2_ids=di)
    poses = rng.normal(0, 1, (5, 7)) * [1e3, 30, 30, 5, 0.8, 0.8, 0.8] + [1.6e9, 0, 0, 0, 0, 0, 0]
    out.update(tum_poses=poses, tum_lines=np.array([orc.format_tum_pose(q[0], q[1:4], q[4:7]) for q in poses]))
    kf_xyz = np.concatenate([clouds[1][0], np.ones((clouds[1][0].shape[0], 1), np.float32)], axis=1)
    nrm, nn = orc.update_normals(kf_xyz, neighbours=True)
    out.update(kf_xyz=kf_xyz, kf_normals=nrm, kf_neighbours=nn)
    np.savez_compressed(os.path.join(HERE, "next_rows_small.npz"), **out)


if __name__ == "__main__":
    if "next" not in sys.argv:
        window_case()
        keyframe_case()
    next_rows_case()
    for f in ("window_small.npz", "keyframes_small.npz", "next_rows_small.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
