import sys
sys.path.insert(0, '.')
import numpy as np
from dmsa_lidar_slam_amd import synth
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
from oracle import oracle_py as orc
prob = synth.window_problem(seed=7, scans=4, rings=32, az_steps=256, num_static=8000)
s = DmsaOptimSettings.sliding_window(num_iter=5)
p_ref, p_gpu = prob.copy(), prob.copy()
rep_ref, gl_ref, tr_ref = orc.optimize_window(p_ref, s, want_global=True)
opt = DmsaOptimizer(pose_table_host=True)
rep = opt.optimizeSet(p_gpu, s)
for a, b in zip(tr_ref, opt.trace()):
    print('ref', a); print('gpu', b)
print(p_ref.relTranslations - p_gpu.relTranslations)
