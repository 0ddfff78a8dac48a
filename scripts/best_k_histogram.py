#!/usr/bin/env python
"""Would speculating on the line search pay?  

adaptiveStepSize (DmsaOptimizer.h:152-182) returns the k in 1..9 whose trial 0.1 k step has the smallest error, or 0 when no trial beats
error0 (the set is then left at raw + 0.9 step, :130-134).  The next iteration voxelises the points of that outcome -- the 205 us stage that
could run BESIDE the nine trial evaluations if the outcome were known in advance.  This script logs best_k of every iteration where the
fixed-iteration bench cannot rig it: whole sequences (every window and keyframe pass of examples/sequence_demo.py, early exits on, the
reference's num_iter) and single calls on the bench shapes with early exits on.

    python scripts/best_k_histogram.py > profiles/r06_best_k_histogram.txt        (needs the GPU: decisions are bit-identical to the oracle's)
"""
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))

import sequence_demo  # noqa: E402
from dmsa_lidar_slam_amd import synth  # noqa: E402
from dmsa_lidar_slam_amd.api import DmsaOptimizer  # noqa: E402
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings  # noqa: E402


class LoggingBackend(sequence_demo.GpuBackend):
    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.window_ks, self.keyframe_ks = [], []

    def optimizeSet(self, prob, settings):
        rep = self.optimizer.optimizeSet(prob, settings)
        self.window_ks.append([t["best_k"] for t in self.optimizer.trace()[: rep.iterations]])
        return rep

    def optimizeKeyframes(self, submap, settings):
        rep = self.kf_optimizer.optimizeSet(submap, settings)
        self.keyframe_ks.append([t["best_k"] for t in self.kf_optimizer.trace()[: rep.iterations]])
        return rep


def show(name, calls):
    flat = [k for c in calls for k in c]
    if not flat:
        print(f"{name}: no iterations")
        return collections.Counter()
    h = collections.Counter(flat)
    n = len(flat)
    top, cnt = h.most_common(1)[0]
    # what a predictor "same k as the previous iteration of this call" would get (the first iteration of a call predicts 9)
    same = sum(1 for c in calls for i, k in enumerate(c) if k == (c[i - 1] if i else 9))
    print(f"{name}: {len(calls)} calls, {n} iterations; best_k histogram " + " ".join(f"{k}:{h.get(k, 0)}" for k in range(10)) +
          f"; most frequent k = {top} ({100.0 * cnt / n:.0f} %); 'same as the previous iteration' would hit {100.0 * same / n:.0f} %")
    return h


def main():
    total = collections.Counter()
    print("# best_k of adaptiveStepSize per iteration, early exits ON (0 = no trial beat error0: the set stays at raw + 0.9 step)")
    seqs = [("sequence ouster 32 x 256, num_iter 15", dict(scans=10, rings=32, az_steps=256, num_iter=15), "ouster"),
            ("sequence + keyframe optimisation", dict(scans=14, rings=32, az_steps=256, num_iter=15, dist_new_keyframe=0.25, num_iter_keyframe_optim=15), "ouster"),
            ("sequence hesai + IMU (config 2 shape)", dict(scans=12, rings=32, az_steps=256, num_iter=15, dist_new_keyframe=0.25, num_iter_keyframe_optim=15, use_imu=True, hesai=True), "hesai"),
            ("sequence livox rosette (config 5 shape)", dict(scans=10, livox=True, num_iter=15, max_points_per_scan=1000, dist_new_keyframe=0.3, num_iter_keyframe_optim=15), "livoxXYZRTLT_ns"),
            ("sequence ouster 64 x 512", dict(scans=12, rings=64, az_steps=512, num_iter=15, dist_new_keyframe=0.25, num_iter_keyframe_optim=15), "ouster")]
    for name, args, sensor in seqs:
        b = LoggingBackend(parity=True, sensor=sensor)
        sequence_demo.run(backend=b, **args)
        total += show(name + " -- windows", b.window_ks)
        total += show(name + " -- keyframe passes", b.keyframe_ks)
    singles = [("config-2 window (5 x 3072 + 10^4 static, IMU rows)", lambda sd: synth.window_problem(seed=sd, scans=5, rings=32, az_steps=96, num_static=10_000, use_imu=True),
                DmsaOptimSettings.sliding_window(use_imu=True)),
               ("rosette window (5 x 24 000 + 2 x 10^4 static)", lambda sd: synth.rosette_window_problem(seed=sd), DmsaOptimSettings.sliding_window()),
               ("bench window (10 x 131 072 + 2 x 10^5 static)", lambda sd: synth.window_problem(seed=sd), DmsaOptimSettings.sliding_window()),
               ("keyframe neighbourhood (32 frames, P = 186)", lambda sd: synth.keyframe_problem(seed=sd, frames=32, arc=2 * np.pi * 32 / 256.0).getSubmap(0, 31),
                DmsaOptimSettings.keyframe_map())]
    for name, make, s in singles:
        calls = []
        for sd in (1, 2, 3):
            opt = DmsaOptimizer(device=0)
            s.num_iter = 15
            rep = opt.optimizeSet(make(sd), s)
            calls.append([t["best_k"] for t in opt.trace()[: rep.iterations]])
            opt.close()
        total += show(name + " -- single calls, seeds 1-3", calls)
    n = sum(total.values())
    print("# all of the above: " + " ".join(f"{k}:{total.get(k, 0)}" for k in range(10)) + f" of {n} iterations; k = 9 in {100.0 * total.get(9, 0) / n:.0f} %, k = 0 in {100.0 * total.get(0, 0) / n:.0f} %")


if __name__ == "__main__":
    main()
