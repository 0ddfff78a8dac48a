#!/usr/bin/env python
"""Does the iteration time depend on where the optimiser converges?  The bench window from several seeds and warm-up lengths: us per iteration of
a 40-iteration call, Gaussians, failed exactness tests.  (Round 6: a fixed point with one ill-placed member in a long Gaussian cost 7 % before the
block-wise fall-back of the latency tier.)   usage (GPU box): python scripts/state_sweep.py [window|rosette|imu]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

kind = sys.argv[1] if len(sys.argv) > 1 else "window"
for seed in range(1, 7):
    if kind == "window":
        prob, s = synth.window_problem(seed=seed), DmsaOptimSettings.sliding_window()
    elif kind == "rosette":
        prob, s = synth.rosette_window_problem(seed=seed, scans=5, pts_per_scan=24000, num_static=20000), DmsaOptimSettings.sliding_window()
    else:
        prob, s = synth.window_problem(seed=seed, scans=5, rings=32, az_steps=96, num_static=10000, use_imu=True), DmsaOptimSettings.sliding_window(use_imu=True)
    row = []
    for warm in (2, 5, 9):
        opt = DmsaOptimizer(fixed_iters=True)
        opt.upload(prob.copy())
        s.num_iter = warm
        opt.optimizeResident(s)
        s.num_iter = 40
        opt.serialFallbackSums(reset=True)
        opt.synchronize()
        t0 = time.perf_counter()
        rep = opt.optimizeResident(s)
        dt = time.perf_counter() - t0
        row.append(f"{1e6 * dt / 40:6.1f} us ({rep.num_gaussians} G, {opt.serialFallbackSums()} fb)")
        opt.close()
    print(f"{kind} seed {seed}: warm-up 2 / 5 / 9 ->", " | ".join(row), flush=True)
