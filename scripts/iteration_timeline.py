#!/usr/bin/env python
"""One iteration of the bench as a kernel timeline (start, duration, end in microseconds relative to the lattice kernel of the iteration;
HIP stream id), from the rocpd database of `rocprofv3 --kernel-trace`:

    python scripts/iteration_timeline.py gpurun_out/r02/stats/stats_results.db [iteration] > profiles/r02_iteration_timeline.txt

The kernel trace keeps the streams' real overlap, but every HIP call of the host is slower under the profiler, so host-bound gaps
(between a read-back and the launches that follow it) are wider than in a plain run."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
it = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
first = [i for i, r in enumerate(rows) if "k_lattice" in r[0]]  # one per voxelisation in every path
i0, i1 = first[it], first[it + 1]
t0 = rows[i0][1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dmsa::", "")
    return n[:44]


streams = sorted({r[3] for r in rows[i0 - 3:i1 - 3]})
print(f"# iteration {it} of the profiled bench run: {(rows[i1][1] - t0) / 1e3:.0f} us from one k_lattice to the next")
print(f"# {'start':>8s} {'dur':>7s} {'end':>8s}  stream  kernel")
for r in rows[i0 - 3:i1 - 3]:
    print(f"  {(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} {(r[2] - t0) / 1e3:8.1f}  s{streams.index(r[3])}      {short(r[0])}")
