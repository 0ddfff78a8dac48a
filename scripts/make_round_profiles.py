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

# 2. PMC summary + HBM traffic per launch pair
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in sorted(glob.glob(os.path.join(src, "p*", "*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        kn = "k_residuals_tiles" if "k_residuals_tiles" in r["Kernel_Name"] else "k_residuals_big"
        # the Jacobian batch (31 evaluations) uses the larger grid of each kernel
        per[kn][int(r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = [f"# rocprofv3 --pmc passes (one pass per counter group, --kernel-trace only) of `python bench.py --steps 3 --warmup 1 --cpu-iters 0`",
         "# workload window10x131072+static200000; means over the dispatches of each launch shape",
         "# FETCH_SIZE / WRITE_SIZE are KiB.  On gfx950 FETCH_SIZE reports half the bytes of a wide coalesced 16 B/lane stream",
         "# (MI355X_MICROARCH.md, HBM): both the raw value and the x2-corrected value are listed; Infinity-Cache hits are included."]
traffic = {}
for kn in sorted(per):
    for shape, ctrs in sorted(per[kn].items(), reverse=True):
        c = {k: sum(v) / len(v) for k, v in ctrs.items()}
        lines.append(f"\n## {kn}: grid_size={shape} threads, dispatches={len(next(iter(ctrs.values())))}")
        for k in sorted(c):
            lines.append(f"  {k:24s} {c[k]:14.5g}")
        if "FETCH_SIZE" in c:
            raw = c["FETCH_SIZE"] * 1024
            wr = c.get("WRITE_SIZE", 0.0) * 1024
            lines.append(f"  -> memory-side read {raw / 1e6:.1f} MB raw / {2 * raw / 1e6:.1f} MB with the gfx950 x2 correction; write {wr / 1e6:.2f} MB per launch")
            traffic.setdefault(kn, []).append((shape, 2 * raw + wr))
        if "TCC_HIT_sum" in c:
            lines.append(f"  -> L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                lines.append(f"  -> {k}/SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES']:.3f}")
open(os.path.join(dst, f"{tag}_pmc_correspondence.txt"), "w").write("\n".join(lines) + "\n")

# one evaluation batch = one k_residuals_tiles launch + one k_residuals_big launch; average the two batch shapes (31 and 9
# evaluations, one of each per iteration) like bench.py averages its launches
def avg(kn):
    v = [t for _, t in traffic.get(kn, [])]
    return sum(v) / len(v) if v else 0.0

bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
tj = {"workload": bench["config"]["workload"], "hbm_bytes_per_launch": round(avg("k_residuals_tiles") + avg("k_residuals_big")),
      "source": f"profiles/{tag}_pmc_correspondence.txt (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024, per evaluation-batch launch pair)"}
open(os.path.join(dst, f"{tag}_traffic.json"), "w").write(json.dumps(tj, indent=1) + "\n")
open(os.path.join(dst, f"{tag}_bench.json"), "w").write(json.dumps(bench) + "\n")
print(open(os.path.join(dst, f"{tag}_kernel_stats.txt")).read()[:2500])
print(tj)
