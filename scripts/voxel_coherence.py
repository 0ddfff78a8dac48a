#!/usr/bin/env python
"""How much of one iteration's voxel order survives into the next?  

Points move by at most max_step (1 cm / 0.01 rad at the control poses) between two iterations of optimizeSet while the voxels measure
2 x and 5 x minGridSize (30 / 75 cm on the bench window), so most points keep their leaf -- IF the lattice stays where it is: PCL anchors
the octree's box at the first finite point (OctreePointCloud::adoptBoundingBoxToPoint), which is a window point and moves too.  With the
debug switch voxel_coherence = 1 the library keeps the leaf codes of every point and counts, per voxelisation and level, how many differ
from the previous voxelisation's (csrc/voxelize_driver.cpp); this script drives one iteration at a time and prints the fractions.

    python scripts/voxel_coherence.py [--out gpurun_out/voxel_coherence.json]        (GPU box)
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, prob, settings, iters):
    from dmsa_lidar_slam_amd.api import DmsaOptimizer

    opt = DmsaOptimizer(fixed_iters=True, debug={"voxel_coherence": 1})
    opt.upload(prob)
    settings.num_iter = 1
    rows, last = [], opt.debugCounters()
    n = prob.localPoints.shape[0] + (prob.staticPoints.shape[0] if hasattr(prob, "staticPoints") else 0)
    for it in range(iters):
        rep = opt.optimizeResident(settings)
        c = opt.debugCounters()
        cmp_, chg, lat = (c[k] - last[k] for k in ("voxel_codes_compared", "voxel_codes_changed", "voxel_lattice_changes"))
        last = c
        rows.append({"iteration": it, "compared": cmp_, "changed": chg, "fraction": (chg / cmp_) if cmp_ else None, "lattice_changes": lat,
                     "step_norm": rep.last_step_norm, "best_k": rep.last_line_search_k})
    opt.close()
    fr = [r["fraction"] for r in rows if r["fraction"] is not None]
    print(f"{name}: points {n}, changed-leaf fraction per iteration (both levels): " + " ".join(f"{100 * f:.1f}%" for f in fr) +
          f" | lattice changes {sum(r['lattice_changes'] for r in rows)}", flush=True)
    return {"points": int(n), "iterations": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--iters", type=int, default=12)
    args = ap.parse_args()
    import numpy as np

    from dmsa_lidar_slam_amd import synth
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

    out = {}
    out["bench_window"] = run("bench window (10 x 131072 + 200000 static)", synth.window_problem(seed=1, scans=10, rings=128, az_steps=1024, num_static=200_000),
                              DmsaOptimSettings.sliding_window(), args.iters)
    out["rosette"] = run("rosette (config 5 shape)", synth.rosette_window_problem(seed=2, scans=5, pts_per_scan=24000, num_static=20000), DmsaOptimSettings.sliding_window(), args.iters)
    full = synth.keyframe_problem(seed=1, frames=32, arc=2 * np.pi * 32 / 256.0)
    out["keyframes_P186"] = run("keyframe neighbourhood (32 frames)", full, DmsaOptimSettings.keyframe_map(), args.iters)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
