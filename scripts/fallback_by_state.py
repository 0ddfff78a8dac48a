import os, sys
sys.path.insert(0, "/root/repo")
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
import time
for warm in (3, 5):
    prob = synth.window_problem(seed=1)
    opt = DmsaOptimizer(fixed_iters=True)
    opt.upload(prob)
    s = DmsaOptimSettings.sliding_window(num_iter=warm)
    opt.optimizeResident(s)
    s.num_iter = 60
    opt.serialFallbackSums(reset=True)
    opt.synchronize()
    t0 = time.perf_counter()
    rep = opt.optimizeResident(s)
    dt = time.perf_counter() - t0
    print("warmup", warm, "fallback sums in 60 iterations:", opt.serialFallbackSums(), "Gaussians", rep.num_gaussians, "us/iter", 1e6 * dt / 60)
    opt.close()
