"""Generates the golden fixtures in this directory.

The reference ships no tests or golden vectors for this path and cannot be built here, so these
fixtures are INPUT/OUTPUT DATA produced by the CPU oracle (oracle/dmsa_oracle.cpp) at the commit that introduced them:
they pin the oracle against regressions (tests/test_golden.py, CPU) and give the GPU parity tests expected outputs
that do not depend on a live oracle build.  Third-party pieces of the oracle are cross-checked independently against
scipy / an explicit PCL-octree model in tests/test_oracle_math.py and tests/test_oracle_voxel.py.

    python tests/golden/make_golden.py        # rewrites window_small.npz / keyframes_small.npz
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


if __name__ == "__main__":
    window_case()
    keyframe_case()
    for f in ("window_small.npz", "keyframes_small.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
