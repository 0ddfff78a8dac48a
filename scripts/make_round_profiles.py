#!/usr/bin/env python
"""Build the committed profiles/ summaries of a round from gpurun_out/<round>/ (written by scripts/profile_round.sh).

    python scripts/make_round_profiles.py r01
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

# 1. kernel stats
out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), "stats", os.path.join(src, "stats", "stats_results.db")])
open(os.path.join(dst, f"{tag}_kernel_stats.txt"), "wb").write(out)

# 2. PMC summary + HBM traffic per evaluation batch
def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n[n.index("k_"):] if "k_" in n else n


per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
rows = []
for f in sorted(glob.glob(os.path.join(src, "p*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        # the Jacobian batch (P + 1 evaluations) and the line-search batch (9 evaluations) differ in workgroup count
        rows.append((short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]), r["Counter_Name"], float(r["Counter_Value"])))
# the workgroup count follows the Gaussian count of the iteration; bucket every kernel's launches into its two batch kinds
lohi = {}
for kn, grid, wg, _, _ in rows:
    lo, hi = lohi.get(kn, (grid, grid))
    lohi[kn] = (min(lo, grid), max(hi, grid))
wgs = {}
for kn, grid, wg, cn, v in rows:
    lo, hi = lohi[kn]
    kind = "Jacobian batch (P + 1 evaluations)" if (hi > 1.5 * lo and grid > (lo + hi) / 2) else "line-search batch (9 evaluations)" if hi > 1.5 * lo else "all launches"
    per[kn][kind][cn].append(v)
    wgs.setdefault((kn, kind), []).append(grid // wg)
    wgs[(kn, kind, "wg")] = wg
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
lines = [f"# rocprofv3 --pmc passes (one pass per counter group, --kernel-trace only) of `python bench.py --steps 3 --warmup 1 --cpu-iters 0`",
         f"# workload {bench['config']['workload']}; means over the dispatches of each launch shape",
         "# FETCH_SIZE / WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced 16 B/lane stream",
         "# (MI355X_MICROARCH.md, HBM): both the raw value and the x2-corrected value are listed; Infinity-Cache hits are included."]
traffic = {}
for kn in sorted(per):
    for kind, ctrs in sorted(per[kn].items()):
        c = {k: sum(v) / len(v) for k, v in ctrs.items()}
        w = wgs[(kn, kind)]
        lines.append(f"\n## {kn}, {kind}: workgroups={min(w)}..{max(w)} x {wgs[(kn, kind, 'wg')]} threads, dispatches={len(next(iter(ctrs.values())))}")
        for k in sorted(c):
            lines.append(f"  {k:24s} {c[k]:14.5g}")
        if "FETCH_SIZE" in c:
            raw = c["FETCH_SIZE"] * 1024
            wr = c.get("WRITE_SIZE", 0.0) * 1024
            lines.append(f"  -> memory-side read {raw / 1e6:.1f} MB raw / {2 * raw / 1e6:.1f} MB with the gfx950 x2 correction; write {wr / 1e6:.2f} MB per launch")
            traffic[f"{kn} / {kind}"] = 2 * raw + wr
        if "TCC_HIT_sum" in c:
            lines.append(f"  -> L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if k in c:
                    lines.append(f"  -> {k}/SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
            lines.append(f"  -> VALU instructions per wave {c['SQ_INSTS_VALU'] / c['SQ_WAVES']:.0f}, LDS {c.get('SQ_INSTS_LDS', 0) / c['SQ_WAVES']:.0f}")
open(os.path.join(dst, f"{tag}_pmc_correspondence.txt"), "w").write("\n".join(lines) + "\n")

# an iteration runs two evaluation batches (P + 1 evaluations, then 9), each one launch of every kernel family (the lane-per-
# evaluation kernel is instantiated per batch width); bench.py averages its event pairs over both, so: all launches / 2
total = sum(traffic.values()) / 2
path = "fast_sums" if "fast" in bench["config"]["path"] else "default"
tj = {"workload": bench["config"]["workload"], "path": path, "hbm_bytes_per_launch": round(total),
      "kernel_launches": {k: round(v) for k, v in traffic.items()},
      "source": f"profiles/{tag}_pmc_correspondence.txt (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, all kernels of the two evaluation batches of an iteration / 2)"}
open(os.path.join(dst, f"{tag}_traffic.json"), "w").write(json.dumps(tj, indent=1) + "\n")
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(json.dumps(bench) + "\n")
# 3. keyframe-set kernel stats and the repeated bench lines, when the round script collected them
kf = os.path.join(src, "kfstats", "stats_results.db")
if os.path.exists(kf):
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), "stats", kf])
    open(os.path.join(dst, f"{tag}_keyframes_kernel_stats.txt"), "wb").write(out)
runs = os.path.join(src, "bench_runs.jsonl")
if os.path.exists(runs):
    rr = [json.loads(l) for l in open(runs) if l.strip().startswith("{")]
    keep = [{k: r[k] for k in ("value", "ms_per_step", "steps", "stage_ms_per_step") if k in r} | {"roofline_avg_launch_ms": r["roofline"]["avg_launch_ms"]} for r in rr]
    open(os.path.join(dst, f"{tag}_bench_runs.json"), "w").write(json.dumps(keep, indent=1) + "\n")
# 3b. HBM traffic of the correspondence kernels on one keyframe neighbourhood (scripts/kf_pmc.sh: kp3 = FETCH_SIZE, kp4 = WRITE_SIZE)
kf_rows = []
for f in sorted(glob.glob(os.path.join(src, "kp*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        kf_rows.append((short(r["Kernel_Name"]), int(r["Dispatch_Id"]) if "Dispatch_Id" in r else 0, r["Counter_Name"], float(r["Counter_Value"])))
if kf_rows:
    tot = collections.defaultdict(lambda: collections.defaultdict(list))
    for kn, _, cn, v in kf_rows:
        tot[kn][cn].append(v)
    # every iteration launches each chain tier twice (Jacobian batch, line-search batch) and the lane-per-evaluation kernel once per batch
    # width: sum of the mean bytes of a launch over ALL launches of the run / number of evaluation batches
    n_batches = len(tot["k_residuals_chain<4, false, 32>"]["FETCH_SIZE"])
    rd = sum(sum(c["FETCH_SIZE"]) for c in tot.values()) * 1024 * 2
    wr = sum(sum(c.get("WRITE_SIZE", [0.0])) for c in tot.values()) * 1024
    nb_w = len(tot["k_residuals_chain<4, false, 32>"].get("WRITE_SIZE", [])) or n_batches
    per_launch = rd / n_batches + wr / nb_w
    lines = ["# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of `python bench.py --workload keyframes --map-frames 0 --frames 32 --steps 3 --warmup 1`",
             "# one neighbourhood of 32 keyframes, P = 186: evaluation batches of 187 and 9; KiB per dispatch, means over the dispatches of a kernel"]
    for kn in sorted(tot):
        c = tot[kn]
        f_ = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        w_ = sum(c.get("WRITE_SIZE", [0.0])) / max(1, len(c.get("WRITE_SIZE", [0.0])))
        lines.append(f"{kn:40s} dispatches {len(c['FETCH_SIZE']):3d}   FETCH_SIZE {f_:10.1f} KiB (x2 on gfx950: {2 * f_ * 1024 / 1e6:7.2f} MB)   WRITE_SIZE {w_:10.1f} KiB ({w_ * 1024 / 1e6:7.2f} MB)")
    lines.append(f"# per evaluation batch (all tiers): {per_launch / 1e6:.1f} MB")
    lines.append("# k_residuals_small<64> writes ~48 MB per Jacobian batch for ~12 MB of residuals: lane = evaluation, so the 8-byte stores of a wave go to 64")
    lines.append("# different columns of E (32-byte write granules); E is indexed by Gaussian because the normal equations sum its rows in that order.")
    open(os.path.join(dst, f"{tag}_pmc_keyframes.txt"), "w").write("\n".join(lines) + "\n")
    open(os.path.join(dst, f"{tag}_traffic_keyframes.json"), "w").write(json.dumps(
        {"workload": "keyframes: one neighbourhood of 32 frames, P = 186", "hbm_bytes_per_launch": round(per_launch),
         "source": f"profiles/{tag}_pmc_keyframes.txt (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, all correspondence kernels / evaluation batches)"}, indent=1) + "\n")
# 4. files the round script wrote in their final form: timelines, the keyframe workload's own line, the driver's command, small windows, gap stamps
import shutil
for name, out_name in (("iteration_timeline.txt", "iteration_timeline.txt"), ("keyframes_iteration_timeline.txt", "keyframes_iteration_timeline.txt"),
                       ("small_window_imu.txt", "small_window_imu.txt"), ("small_window_rosette.txt", "small_window_rosette.txt")):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, f"{tag}_{out_name}"))
kfb = os.path.join(src, "bench_keyframes.json")
if os.path.exists(kfb):
    lines = [l for l in open(kfb).read().splitlines() if l.startswith("{")]
    if lines:
        open(os.path.join(dst, f"{tag}_bench_keyframes.json"), "w").write(lines[-1] + "\n")
drv = os.path.join(src, "bench_driver_command.jsonl")
if os.path.exists(drv):
    rr = [json.loads(l) for l in open(drv) if l.strip().startswith("{")]
    open(os.path.join(dst, f"{tag}_bench_driver_command.json"), "w").write(json.dumps([{k: r[k] for k in ("value", "ms_per_step", "steps", "warmup")} for r in rr], indent=1) + "\n")
gs = os.path.join(src, "gap_stamps.txt")
if os.path.exists(gs) and os.path.getsize(gs) > 0:
    head = ("# DMSA_DEBUG=gap_stamps=1: one-thread kernels write the device's 100 MHz wall clock in front of, between and behind the kernels of the\n"
            "# normal equations and the LM solve (unprofiled runs, last iteration of a call).  The rocprofv3 kernel trace of round 4\n"
            "# (profiles/r04_keyframes_iteration_timeline.txt) shows three ~40 us holes between k_sync_wait -> k_jacobian_columns -> k_normal_eq_mfma ->\n"
            "# k_normal_eq_reduce; the kernels themselves take 20 + 25 + 9 = 54 us.  Without the profiler slowing the host's launches:\n")
    tail = ("# ~55-60 us for 54 us of kernels (+ the stamp kernels' own ~3 us each): the holes are the profiler's (every HIP launch costs tens of\n"
            "# microseconds of host time under rocprofv3, and at that point of the iteration the host enqueues just in time).  They do not exist in a plain run.\n")
    open(os.path.join(dst, f"{tag}_keyframes_gap_stamps.txt"), "w").write(head + open(gs).read() + tail)
print(open(os.path.join(dst, f"{tag}_kernel_stats.txt")).read()[:2500])
print(tj)
