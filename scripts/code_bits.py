"""Leaf code widths (bits) of the two voxel levels on the bench shapes: how many 8-bit passes the radix sort needs."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
for name, prob, s in (("window", synth.window_problem(seed=1), DmsaOptimSettings.sliding_window(num_iter=1)),
                      ("small_imu", synth.window_problem(seed=5, scans=5, rings=32, az_steps=96, num_static=10_000, use_imu=True), DmsaOptimSettings.sliding_window(use_imu=True, num_iter=1)),
                      ("kf32", synth.keyframe_problem(seed=1, frames=32, arc=2 * np.pi * 32 / 256.0), DmsaOptimSettings.keyframe_map(num_iter=1))):
    opt = DmsaOptimizer()
    opt.upload(prob)
    opt.poseTables(prob.getPoseParameters())
    opt.updateGlobalPoints(0)
    opt.buildGaussians(s)
    for l in (0, 1):
        info, code, key, order = opt.voxelLevel(l)
        valid = code[order] if len(order) else code
        print(name, "level", l, "max code bits", int(valid.max()).bit_length(), "distinct leaves", len(np.unique(valid)), {f: getattr(info, f) for f, _ in info._fields_})
    opt.close()
