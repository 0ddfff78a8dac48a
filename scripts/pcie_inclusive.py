"""PCIe-inclusive rate of the drop-in call: optimizeSet with host buffers (upload + N iterations + pose read-back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

prob = synth.window_problem(seed=1)
opt = DmsaOptimizer(device=0, fixed_iters=True)
for iters in (10, 10, 10):
    s = DmsaOptimSettings.sliding_window(num_iter=iters)
    p = prob.copy()
    t0 = time.perf_counter()
    rep = opt.optimizeSet(p, s)
    dt = time.perf_counter() - t0
    print(f"optimizeSet(host buffers), {rep.iterations} iterations: {1e3 * dt:.2f} ms -> {rep.iterations / dt:.1f} it/s")
