"""Timing of the addStaticPoints step at the bench size: GPU (host buffers in, results out) vs the CPU oracle."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmsa_lidar_slam_amd import synth  # noqa: E402
from dmsa_lidar_slam_amd.static_points import StaticPointSelector  # noqa: E402

p = synth.static_select_problem(seed=1)
g = StaticPointSelector(0)
f32 = np.float32
half = f32(p.minGridSize) / f32(2.0)


def timeit(fn, reps=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps * 1e3, out


t_sel, sel = timeit(lambda: g.selectStaticPoints(p))
t_ds, pick = timeit(lambda: g.randomGridDownsampling(sel.staticPoints, half, 7))
active = sel.staticPoints[pick]
t_ov, ov = timeit(lambda: g.getOverlap(active, p.windowPoints, p.minGridSize))
t_all, _ = timeit(lambda: g.addStaticPoints(p, 7))
out = {"window_points": int(p.windowPoints.shape[0]), "keyframe_points": int(p.keyPoints.shape[0]), "selected": int(sel.staticPoints.shape[0]),
       "active": int(active.shape[0]), "overlap": ov[0], "gpu_ms": {"select": round(t_sel, 3), "thin": round(t_ds, 3), "overlap": round(t_ov, 3), "addStaticPoints": round(t_all, 3)}}
# the same with the window cloud RESIDENT in the optimizer's context (NULL window pointers): identity poses make global == local
from dmsa_lidar_slam_amd.api import DmsaOptimizer  # noqa: E402
from dmsa_lidar_slam_amd.keyframe_cloud import KeyframeCloudBuilder  # noqa: E402
from dmsa_lidar_slam_amd.problems import ContinuousTrajectory  # noqa: E402
from dmsa_lidar_slam_amd.static_points import StaticSelectProblem  # noqa: E402

nw = p.windowPoints.shape[0]
ident = ContinuousTrajectory(relOrientations=np.zeros((3, 3)), relTranslations=np.zeros((3, 3)), stamps=np.array([0.0, 0.5, 1.0]), trajTime=np.linspace(0.0, 1.0, 11),
                             localPoints=p.windowPoints, tformIdPerPoint=np.zeros(nw, np.int32), ringIds=np.zeros(nw, np.int32), minGridSize=p.minGridSize)
opt = DmsaOptimizer(device=0)
opt.upload(ident)
opt.poseTables(opt.getPoseParameters(), download=False)
opt.updateGlobalPoints(0, download=False)
gr = StaticPointSelector(optimizer=opt)
pr = StaticSelectProblem(windowPoints=None, numWindowResident=nw, keyframeIds=p.keyframeIds, frameOffsets=p.frameOffsets, keyPoints=p.keyPoints, keyNormals=p.keyNormals,
                         keyRingIds=p.keyRingIds, currPos=p.currPos, minGridSize=p.minGridSize)
t_res, res_out = timeit(lambda: gr.addStaticPoints(pr, 7))
assert np.array_equal(res_out[1], active)
kr = KeyframeCloudBuilder(optimizer=opt)
t_kfr, _ = timeit(lambda: kr.addNewKeyframeCloud(None, None, p.minGridSize, 7, np.zeros(3), np.zeros(3), numResident=nw), 5)
out["resident_window_gpu_ms"] = {"addStaticPoints": round(t_res, 3), "keyframeCloud": round(t_kfr, 3)}
# DmsaSlam::preProcess of one raw 128 x 1024 scan (rays onto a 50 x 36 x 6 m box), Config.h defaults
rng = np.random.default_rng(5)
d = rng.normal(size=(131072, 3))
d[:, 2] *= 0.3
d /= np.linalg.norm(d, axis=1, keepdims=True)
raw = np.concatenate([d * np.min(np.array([25.0, 18.0, 3.0])[None, :] / np.maximum(np.abs(d), 1e-9), axis=1)[:, None], np.ones((131072, 1))], axis=1).astype(f32)
t_pre, pre = timeit(lambda: g.preProcess(raw, 7))
out["preProcess"] = {"raw_points": int(raw.shape[0]), "filtered": int(pre[0].shape[0]), "grid": pre[2], "gpu_ms": round(t_pre, 3)}
# keyframe creation (addNewKeyframeToMap :497-531): thinning + local frame + normals (k = 6) of one window's cloud
from dmsa_lidar_slam_amd.keyframe_cloud import KeyframeCloudBuilder  # noqa: E402

kb = KeyframeCloudBuilder(0)
ids = np.zeros(p.windowPoints.shape[0], np.int32)
pos0, orient0 = np.array([0.1, -0.2, 0.05]), np.array([0.01, -0.02, 0.3])
t_kf, kf = timeit(lambda: kb.addNewKeyframeCloud(p.windowPoints, ids, p.minGridSize, 7, pos0, orient0), 5)
t_nrm, _ = timeit(lambda: kb.updateNormals(kf[0], 2 * p.minGridSize), 5)
out["keyframeCloud"] = {"global_points": int(p.windowPoints.shape[0]), "keyframe_points": int(kf[0].shape[0]), "gpu_ms": round(t_kf, 3), "normals_only_gpu_ms": round(t_nrm, 3)}
if "--cpu" in sys.argv:
    from oracle import oracle_py as orc

    c_sel, rs = timeit(lambda: orc.select_static_points(p), 3)
    c_ds, rp = timeit(lambda: orc.random_grid_downsampling(rs.staticPoints, half, 7), 3)
    c_ov, _ = timeit(lambda: orc.get_overlap(rs.staticPoints[rp], p.windowPoints, p.minGridSize), 3)
    out["cpu_oracle_ms"] = {"select": round(c_sel, 2), "thin": round(c_ds, 2), "overlap": round(c_ov, 2)}
    c_pre, rpre = timeit(lambda: orc.preprocess_scan(raw, 7), 3)
    assert np.array_equal(rpre[0], pre[0]) and np.array_equal(rpre[1], pre[1])
    out["preProcess"]["cpu_oracle_ms"] = round(c_pre, 2)
    orc.set_threads(1)
    sub = kf[0][:20000]  # the oracle's neighbour search is exhaustive (O(n^2)): time a 20 000-point subset, one thread
    c_nrm, _ = timeit(lambda: orc.update_normals(sub), 1)
    out["keyframeCloud"]["cpu_oracle_normals_20000_points_ms"] = round(c_nrm, 1)
print(json.dumps(out))
