#!/usr/bin/env python
"""Turn rocprofv3 output (rocpd sqlite db from --kernel-trace --stats, and/or --pmc counter CSVs) into the small text
summaries committed under profiles/.

    python scripts/summarize_profile.py stats gpurun_out/prof1/r1_results.db  > profiles/r01_kernel_stats.txt
    python scripts/summarize_profile.py pmc   gpurun_out/pmc                  > profiles/r01_pmc_k_residuals.txt
"""
import collections
import csv
import glob
import os
import sqlite3
import sys


def stats(db_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    # k_sync_wait is a one-wave kernel that SPINS until another stream has caught up (csrc/dev_sync.h): its "duration" is idle time of a
    # stream, not work.  It is listed, but the pct column is taken over the kernels that compute.
    def waits(name):
        return "k_sync_wait" in name

    tot = sum(r[2] for r in rows if not waits(r[0]))
    print(f"# rocprofv3 --kernel-trace --stats summary of {os.path.basename(db_path)} (durations in us)")
    print(f"# total kernel time {tot / 1e3:.1f} us over {sum(r[1] for r in rows if not waits(r[0]))} dispatches "
          f"(+ {sum(r[2] for r in rows if waits(r[0])) / 1e3:.1f} us of stream waits in {sum(r[1] for r in rows if waits(r[0]))} k_sync_wait dispatches, not in pct)")
    print(f"{'kernel':72s} {'calls':>6s} {'total_us':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for name, n, total, avg, mn, mx in rows:
        short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if "rocprim" in short:
            short = "rocprim::" + short.split("::")[-1][:48] + "<...>"
        pct = "  wait" if waits(name) else f"{100 * total / tot:6.2f}"
        print(f"{short[:72]:72s} {n:6d} {total / 1e3:10.1f} {avg / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {pct}")
    # the correspondence kernels by batch kind: an iteration launches every kernel family twice (P + 1 evaluations for the
    # Jacobian, 9 for the line search); the workgroup count follows the iteration's Gaussian count, so launches are bucketed
    # at the midpoint of each kernel's workgroup range (grid_x is in THREADS in rocprofv3's tables)
    print("\n# correspondence kernels by batch kind (workgroups = grid_x / workgroup size, x grid_y)")
    rows = cur.execute("select name, grid_x, grid_y, workgroup_x, end-start from kernels where name like '%k_residuals%'").fetchall()
    by = collections.defaultdict(list)
    for name, gx, gy, wx, d in rows:
        by[name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")].append(((gx // max(1, wx)) * gy, wx, d))
    for short in sorted(by):
        lo, hi = min(w for w, _, _ in by[short]), max(w for w, _, _ in by[short])
        two = hi > 1.5 * lo
        for kind in (("P+1 evaluations", "9 evaluations") if two else ("all launches",)):
            sel = [(w, wx, d) for w, wx, d in by[short] if not two or (w > (lo + hi) / 2) == (kind == "P+1 evaluations")]
            ds = [d for _, _, d in sel]
            ws = [w for w, _, _ in sel]
            print(f"  {short[:44]:44s} {kind:16s} workgroups={min(ws):6d}..{max(ws):<6d} wg_size={sel[0][1]:4d} launches={len(sel):4d} "
                  f"avg_us={sum(ds) / len(ds) / 1e3:9.2f} min_us={min(ds) / 1e3:9.2f} max_us={max(ds) / 1e3:9.2f}")


def pmc(root):
    print(f"# rocprofv3 --pmc passes for kernels matching k_residuals ({root}); one pass per counter group, kernel-trace only")
    per_shape = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "*", "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            per_shape[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for shape, ctrs in per_shape.items():
        print(f"\n## dispatch grid_size={shape} threads (workgroup 512)")
        for name in sorted(ctrs):
            v = ctrs[name]
            print(f"  {name:24s} mean={sum(v) / len(v):14.5g}  n={len(v)}")
        c = {k: sum(v) / len(v) for k, v in ctrs.items()}
        if "FETCH_SIZE" in c:
            # FETCH_SIZE is in KiB; on gfx950 it reports half of a wide coalesced stream (MI355X_MICROARCH.md, HBM) -> x2
            print(f"  -> HBM read traffic ~ {c['FETCH_SIZE'] * 1024 * 2 / 1e6:.1f} MB per launch (FETCH_SIZE x 1024 x 2, gfx950 correction)")
        if "WRITE_SIZE" in c:
            print(f"  -> HBM write traffic ~ {c['WRITE_SIZE'] * 1024 / 1e6:.2f} MB per launch (uncalibrated)")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            print(f"  -> L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        if "SQ_WAVE_CYCLES" in c:
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if k in c:
                    print(f"  -> {k}/SQ_WAVE_CYCLES = {c[k] / c['SQ_WAVE_CYCLES']:.3f}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
