"""Time the host LM solve of the product (dmsa_lm_solve) at the keyframe-pass size for a few thread counts."""
import ctypes as C
import time

import numpy as np
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dmsa_lidar_slam_amd import _capi as capi

lib = capi.load_library()
for P in (30, 186, 372):
    rng = np.random.default_rng(0)
    A = rng.normal(size=(2 * P + 100, P))
    H = A.T @ A + 1e-5 * np.eye(P)
    g = rng.normal(size=P)
    ref = None
    for th in (1, 4, 8, 16):
        step = np.zeros(P)
        t = time.perf_counter()
        for _ in range(10):
            lib.dmsa_lm_solve(capi.ptr(H, C.c_double), capi.ptr(g, C.c_double), P, 0.2, th, capi.ptr(step, C.c_double))
        dt = (time.perf_counter() - t) / 10 * 1e3
        ref = step if ref is None else ref
        print(f"P={P} threads={th}: {dt:.3f} ms  identical={np.array_equal(step, ref)}")
