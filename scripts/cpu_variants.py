"""CPU timing table of SURVEY.md 8(d): the oracle built at the reference's own -O1 (CMakeLists.txt:15), at -O2 (the shipped test
build) and at -O3 -march=native, one thread, plus the evaluation-parallel OpenMP variant -- all on the bench's window workload,
on the machine this runs on.  Every build is timed in its own process (the library is loaded once per process).

    python scripts/cpu_variants.py [--iters 3]  ->  one JSON line
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(iters, threads):
    from dmsa_lidar_slam_amd import synth
    from dmsa_lidar_slam_amd.problems import DmsaOptimSettings
    from oracle import oracle_py as orc

    prob = synth.window_problem(seed=1)
    s = DmsaOptimSettings.sliding_window(use_imu=False, num_iter=iters)
    orc.set_threads(threads)
    t0 = time.perf_counter()
    rep, _, _ = orc.optimize_window(prob, s, fixed_iters=True)
    dt = time.perf_counter() - t0
    print(json.dumps({"iterations": rep.iterations, "seconds": round(dt, 3), "it_per_s": round(rep.iterations / dt, 4), "threads": threads,
                      "error0": rep.error0}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--child", type=int, default=0)
    args = ap.parse_args()
    if args.child:
        return child(args.iters, args.child)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "all", "variants"])
    builds = {"O1 (the reference's level)": "_variants/libdmsa_oracle_O1.so", "O2 (shipped oracle)": "libdmsa_oracle.so",
              "O3 -march=native": "_variants/libdmsa_oracle_O3native.so"}
    threads = max(1, min(32, os.cpu_count() or 1))
    out = {"workload": "10 x 131072-point window + 200000 static points (bench.py default)", "nproc": os.cpu_count(), "cpu": cpu_model(), "rows": []}
    for name, rel in builds.items():
        for th in ((1, threads) if name.startswith("O2") or name.startswith("O3") else (1,)):
            env = dict(os.environ, DMSA_ORACLE_LIB=os.path.join(ROOT, "oracle", rel))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--iters", str(args.iters), "--child", str(th)], env=env, capture_output=True, text=True, check=True)
            row = json.loads(r.stdout.strip().splitlines()[-1])
            row["build"] = name
            out["rows"].append(row)
    print(json.dumps(out))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
