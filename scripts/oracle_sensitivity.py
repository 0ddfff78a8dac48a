#!/usr/bin/env python
"""Which hypotheses does the 1e-4 m / 1e-4 rad claim depend on?

The reference cannot be built here, so wherever its arithmetic is not spelled out in its own sources the oracle STATES an order
(DESIGN.md section 5).  oracle/dmsa_oracle.cpp carries one compile-time switch per such statement (see its header); this script
builds one oracle per switch (make -C oracle hypotheses), runs the same problems through every build for the same number of
iterations (early exits off) and reports how far the optimised GLOBAL poses move away from the default oracle's.

    python scripts/oracle_sensitivity.py [--iters 5] [--out profiles/r06_oracle_sensitivity.json]

CPU only; about two minutes (the P = 186 keyframe case dominates).  Test infrastructure: nothing here touches the product library.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HYPOTHESES = {
    "TRANSFORM_PAIRWISE": "Matrix4f*Vector4f as (c0x+c1y)+(c2z+c3) instead of ((c0x+c1y)+c2z)+c3 (ContinuousTrajectory.h:151, MapManagement.h:142)",
    "SUM3_LEFT": "3-term fixed-size redux (x0+x1)+x2 instead of x0+(x1+x2) (DmsaOptimizer.h:263, Gaussians.h:52-75)",
    "MAHA_ASSOC": "w*((d^T A) d) instead of ((w d^T) A) d (DmsaOptimizer.h:263)",
    "FIT_FLOAT": "every fit sum (column means, centred products, weight mean) as a scalar float chain in member order (Gaussians.h:146-154, :176)",
    "FIT_MEAN_TREE": "colwise().mean() / VectorXf::mean() as 64-wide trees in double (the statement until round 4) instead of Eigen's own linear-redux float order",
    "FIT_COV_TREE": "centered^T*centered as 64-wide pairwise trees in double, divided in double (the statement until round 5) instead of the float order of "
                    "Eigen 3.4's product kernels (lazy product below 14 members, gebp scalar chains in depth blocks for a 32 KB L1, float division) (Gaussians.h:147)",
    "WEIGHT_DIV": "pow(-1) of the member counts as the correctly rounded 1.0f / n instead of libm's powf(n, -1.0f) (Gaussians.h:172)",
    "LIMITCOV_VT": "limitCovariance rebuilds V*D*V^T instead of V*D*V^-1 with the cofactor inverse (Gaussians.h:200)",
    "LIMITCOV_JACOBI": "limitCovariance's eigenpairs from a fixed 6-sweep cyclic float Jacobi iteration (the statement of rounds 1-5, 'H5') instead of "
                       "EigenSolver<Matrix3f> restated from Eigen 3.4.0 (Hessenberg + Francis QR + back substitution, oracle/eigensolver3f.h) (Gaussians.h:184-188)",
    "EIG_NORMALIZE_SCALAR": "eigenvectors()'s normalize() dividing every row as re/nrm instead of Eigen's Packet2cf division (re*nrm)/(nrm*nrm) in rows 0 and 1 (scalar complex division in row 2)",
    "EIG_BACK_HALVES": "the one 3-term sum inside that solver (back transformation of the last eigenvector, EigenSolver.h doComputeEigenvectors) as x0+(x1+x2) instead of (x0+x1)+x2",
    "STEP_LEFT_ASSOC": "the LM step as Eigen associates it, ((-alpha H^-1) J^T) e with a P x rows temporary and the GEMV's 16-column blocks, instead of (-alpha H^-1)(J^T e) (DmsaOptimizer.h:113)",
    "LM_BLOCKED_LU": "H^-1 from a right-looking partial-pivot LU in 8-column panels + two triangular solves (the shape of Eigen's PartialPivLU::inverse) instead of Gauss-Jordan on [H | I] (DmsaOptimizer.h:113)",
    "EIGEN_L1_48K": "the same product order for a reference machine with a 48 KB L1d: depth blocks of 1016 instead of 680 members (run-time: orc_set_eigen_l1_bytes)",
    "JTJ_NOFMA": "J^T J / J^T e / e^T e at P > 64 with separate multiply and add instead of the fma chain of v_mfma_f64 (DmsaOptimizer.h:107-113)",
    "GLIBC_TRIG": "sin/cos/acos/atan2 from glibc instead of include/dmsa_detmath.h (helpers.h:24-65)",
}
CASES = ["golden_window", "rosette", "window_imu", "keyframes_P72", "keyframes_P186"]


def make_case(name):
    import numpy as np

    from dmsa_lidar_slam_amd import synth
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings

    if name == "golden_window":  # tests/golden/window_small.npz
        return synth.window_problem(seed=21, scans=3, rings=16, az_steps=160, num_static=2500), DmsaOptimSettings.sliding_window(), True
    if name == "rosette":
        return synth.rosette_window_problem(seed=2, scans=4, pts_per_scan=6000, num_static=3000), DmsaOptimSettings.sliding_window(), True
    if name == "window_imu":
        return synth.window_problem(seed=7, scans=5, rings=16, az_steps=256, num_static=1500, use_imu=True), DmsaOptimSettings.sliding_window(use_imu=True), True
    if name == "keyframes_P72":
        return synth.keyframe_problem(seed=4, frames=13, rings=16, az_steps=96, arc=0.8), DmsaOptimSettings.keyframe_map(), False
    if name == "keyframes_P186":  # one neighbourhood of the sharded pass, as bench.py times it
        full = synth.keyframe_problem(seed=1, frames=32, arc=2 * np.pi * 32 / 256.0)
        sub = full.getSubmap(0, 31)
        from dmsa_lidar_slam_amd import posemath

        sub.truth_global = tuple(np.asarray(v)[:32] for v in posemath.relative2global(*full.truth_relative))
        return sub, DmsaOptimSettings.keyframe_map(), False
    raise ValueError(name)


def worker(case, iters):
    """Runs in a subprocess whose DMSA_ORACLE_LIB selects the oracle build."""
    import numpy as np

    sys.path.insert(0, ROOT)
    from oracle import oracle_py as orc

    prob, s, window = make_case(case)
    s.num_iter = iters
    if os.environ.get("DMSA_ORACLE_L1"):
        orc.set_eigen_l1_bytes(int(os.environ["DMSA_ORACLE_L1"]))
    fn = orc.optimize_window if window else orc.optimize_keyframes
    rep, _, trace = fn(prob, s, fixed_iters=True)
    go, gt = orc.relative2global(prob.relOrientations, prob.relTranslations)
    # distance of the optimised translations to the generating trajectory: says whether a variant lands somewhere else or only
    # realises the noise of the numeric Jacobian differently
    if hasattr(prob, "truth_global"):
        truth_t = np.asarray(prob.truth_global[1])
    else:
        truth_t = np.asarray(orc.relative2global(*prob.truth_relative)[1])
    print(json.dumps({"go": np.asarray(go).tolist(), "gt": np.asarray(gt).tolist(), "iterations": int(rep.iterations),
                      "max_abs_dt_to_truth_m": float(np.abs(np.asarray(gt) - truth_t).max()),
                      "structure": [[int(t["M"]), int(t["Mm"]), int(t["best_k"])] for t in trace[: rep.iterations]]}))


def run(case, lib, iters, l1=None):
    env = dict(os.environ)
    env.pop("DMSA_ORACLE_L1", None)
    if l1:
        env["DMSA_ORACLE_L1"] = str(l1)
    if lib:
        env["DMSA_ORACLE_LIB"] = lib
    else:
        env.pop("DMSA_ORACLE_LIB", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", case, "--iters", str(iters)], capture_output=True, text=True, env=env, check=True)
    return json.loads(r.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--hypotheses", default=",".join(HYPOTHESES))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.worker:
        return worker(args.worker, args.iters)
    import numpy as np

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j4", "all", "hypotheses"])
    cases = [c for c in args.cases.split(",") if c]
    hyps = [h for h in args.hypotheses.split(",") if h]
    table = {}
    for case in cases:
        base = run(case, None, args.iters)
        row = {"default": {"max_abs_dt_to_truth_m": base["max_abs_dt_to_truth_m"]}}
        for h in hyps:
            v = run(case, None, args.iters, l1=48 * 1024) if h == "EIGEN_L1_48K" else run(case, os.path.join(ROOT, "oracle", "_variants", f"libdmsa_oracle_{h}.so"), args.iters)
            row[h] = {"max_abs_dt_m": float(np.abs(np.array(v["gt"]) - np.array(base["gt"])).max()),
                      "max_abs_dr_rad": float(np.abs(np.array(v["go"]) - np.array(base["go"])).max()),
                      "same_structure_and_line_search": v["structure"] == base["structure"], "max_abs_dt_to_truth_m": v["max_abs_dt_to_truth_m"]}
        table[case] = row
        print(case, f"default-to-truth {base['max_abs_dt_to_truth_m']:.1e}",
              {h: f"{r['max_abs_dt_m']:.1e}/{r['max_abs_dr_rad']:.1e} truth {r['max_abs_dt_to_truth_m']:.1e}" + ("" if r["same_structure_and_line_search"] else " (*)")
               for h, r in row.items() if h != "default"}, flush=True)
    out = {"iterations": args.iters, "early_exits": "off", "measure": "max |global pose - default oracle's| after the same iteration count (m / rad)",
           "hypotheses": {h: HYPOTHESES[h] for h in hyps}, "cases": table}
    print("\n| hypothesis | " + " | ".join(cases) + " |")
    print("|---|" + "---|" * len(cases))
    print("| (default oracle, distance to the generating trajectory) | " + " | ".join(f"{table[c]['default']['max_abs_dt_to_truth_m']:.1e}" for c in cases) + " |")
    for h in hyps:
        cells = []
        for c in cases:
            r = table[c][h]
            cells.append(f"{max(r['max_abs_dt_m'], r['max_abs_dr_rad']):.1e}" + ("" if r["same_structure_and_line_search"] else " (*)"))
        print(f"| `{h}` | " + " | ".join(cells) + " |")
    print("(*) Gaussian counts or line-search decisions differ from the default oracle's in some iteration")
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
