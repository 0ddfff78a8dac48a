"""Debug: per-phase cycle counts of k_residuals_tiles (needs a -DDMSA_PHASE_CLOCKS build of libdmsa_hip.so)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import numpy as np
from dmsa_lidar_slam_amd import synth, _capi as capi
from dmsa_lidar_slam_amd.api import DmsaOptimizer
from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
prob = synth.window_problem(seed=1)
opt = DmsaOptimizer(fixed_iters=True)
opt.upload(prob)
s = DmsaOptimSettings.sliding_window(num_iter=1)
opt.optimizeResident(s)
lib = capi.load_library()
cap = 8192
lib.dmsa_debug_phase_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
lib.dmsa_debug_phase_clocks(opt._ctx, None, cap)
opt.optimizeResident(s)
out = np.zeros((cap, 8, 8), np.int64)
nt = lib.dmsa_debug_phase_clocks(opt._ctx, out.ctypes.data_as(C.POINTER(C.c_longlong)), cap)
out = out[:nt]
act = out[:, :, 1].sum(axis=1) > 0
print('tiles', nt, 'active (kind 0)', act.sum())
names = ['prologue', 'transform', 'scan1+ends', 'stage next tab', 'barrier wait', 'means/out', 'pass2', 'scan2+ends']
w0 = out[act][:, 0, :]
tot = w0.sum(axis=1)
print('wave0 total cycles per WG (last launch = line search, 9 evals in chunks): mean', tot.mean(), 'max', tot.max())
for k, n in enumerate(names):
    print(f'  {n:16s} mean {w0[:, k].mean():10.0f}  share {w0[:, k].sum() / tot.sum():.3f}')
