import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
# determinism over long calls: any race in the device-side dependencies or the fall-back paths shows as a different trajectory
for name, prob, s in (("window", synth.window_problem(seed=1), DmsaOptimSettings.sliding_window(num_iter=600)),
                      ("window_imu", synth.window_problem(seed=7, scans=5, rings=32, az_steps=96, num_static=10000, use_imu=True), DmsaOptimSettings.sliding_window(use_imu=True, num_iter=1500)),
                      ("rosette", synth.rosette_window_problem(seed=2, scans=5, pts_per_scan=24000, num_static=20000), DmsaOptimSettings.sliding_window(num_iter=1000))):
    outs = []
    for rep in range(3):
        p = prob.copy()
        opt = DmsaOptimizer(fixed_iters=True)
        r = opt.optimizeSet(p, s)
        c = opt.debugCounters() if hasattr(opt, "debugCounters") else None
        outs.append((p.relOrientations.copy(), p.relTranslations.copy(), r.iterations, r.num_gaussians, opt.serialFallbackSums()))
        opt.close()
    same = all(np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]) and outs[0][2:4] == o[2:4] for o in outs[1:])
    print(name, "iterations", outs[0][2], "Gaussians", outs[0][3], "fallback sums", [o[4] for o in outs], "identical over 3 runs:", same, flush=True)
    assert same
