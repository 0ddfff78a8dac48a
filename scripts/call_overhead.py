"""Per-call cost of optimizeSet on a resident window, apart from its iterations: wall time of dmsa_optimize_resident for K = 1 ... 32
iterations (best of 5 each, bench window), the line through them, and the library's own timers for one call.
usage (GPU box): python scripts/call_overhead.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmsa_lidar_slam_amd import synth  # noqa: E402
from dmsa_lidar_slam_amd.api import DmsaOptimizer  # noqa: E402
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings  # noqa: E402

prob = synth.window_problem(seed=1, scans=10, rings=128, az_steps=1024, num_static=200_000)
opt = DmsaOptimizer(fixed_iters=True)
opt.upload(prob)
s = DmsaOptimSettings.sliding_window(num_iter=3)
opt.optimizeResident(s)
ks, ts = [1, 2, 4, 8, 16, 32], []
for k in ks:
    s.num_iter = k
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        opt.optimizeResident(s)
        best = min(best, time.perf_counter() - t0)
    ts.append(1e3 * best)
    print(f"K = {k:2d}: {1e3 * best:8.3f} ms per call, {1e3 * best / k:.4f} ms per iteration")
a, b = np.polyfit(ks, ts, 1)
print(f"fit: {a:.4f} ms per iteration + {b:.4f} ms per call")
# the driver's command shape: a 5-iteration call, then ONE timed 20-iteration call
for rep in range(4):
    s.num_iter = 5
    opt.optimizeResident(s)
    opt.timing(reset=True)
    s.num_iter = 20
    opt.synchronize()
    t0 = time.perf_counter()
    opt.optimizeResident(s)
    opt.synchronize()
    t1 = time.perf_counter()
    print(f"5 then 20 iterations: {1e3 * (t1 - t0) / 20:.4f} ms per iteration")
for rep in range(3):
    t0 = time.perf_counter()
    opt.optimizeResident(s)
    t1 = time.perf_counter()
    print(f"20 again: {1e3 * (t1 - t0) / 20:.4f} ms per iteration")
# what a call that follows a SHORTER call pays: retries?
for rep in range(4):
    s.num_iter = 5
    opt.optimizeResident(s)
    c0 = opt.debugCounters()
    s.num_iter = 20
    opt.synchronize()
    t0 = time.perf_counter()
    r = opt.optimizeResident(s)
    opt.synchronize()
    t1 = time.perf_counter()
    c1 = opt.debugCounters()
    tm = opt.timing()
    print(f"5 then 20: {1e3 * (t1 - t0):.3f} ms, speculation retries {c1['speculation_retries'] - c0['speculation_retries']}, sync retries {c1['sync_retries'] - c0['sync_retries']}, "
          f"iterations {r.iterations}")
