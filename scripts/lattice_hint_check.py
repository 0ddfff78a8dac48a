#!/usr/bin/env python
"""How often do the growth events of the previous voxelisation hold on the bench window (lattice_hint)?  Prints the counters."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
prob = synth.window_problem(seed=1)
for hint in (1, 0):
    opt = DmsaOptimizer(fixed_iters=True, debug={"lattice_hint": hint})
    opt.upload(prob)
    s = DmsaOptimSettings.sliding_window(num_iter=5)
    opt.optimizeResident(s)
    s.num_iter = 100
    t0 = time.perf_counter()
    rep = opt.optimizeResident(s)
    dt = time.perf_counter() - t0
    c = opt.debugCounters()
    print(f"lattice_hint={hint}: {rep.iterations / dt:.1f} it/s, hints held {c['lattice_hints_held']}, replays {c['lattice_replays']}")
    opt.close()
