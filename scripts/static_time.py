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
if "--cpu" in sys.argv:
    from oracle import oracle_py as orc

    c_sel, rs = timeit(lambda: orc.select_static_points(p), 3)
    c_ds, rp = timeit(lambda: orc.random_grid_downsampling(rs.staticPoints, half, 7), 3)
    c_ov, _ = timeit(lambda: orc.get_overlap(rs.staticPoints[rp], p.windowPoints, p.minGridSize), 3)
    out["cpu_oracle_ms"] = {"select": round(c_sel, 2), "thin": round(c_ds, 2), "overlap": round(c_ov, 2)}
print(json.dumps(out))
