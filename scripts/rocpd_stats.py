"""Per-kernel durations out of a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`): count, mean, min, max in us,
sorted by total time.  usage: rocpd_stats.py <results.db> [name filter]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
q = """select s.kernel_name, count(*), avg(d.end - d.start) / 1000.0, min(d.end - d.start) / 1000.0, max(d.end - d.start) / 1000.0, sum(d.end - d.start) / 1000.0
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 6 desc"""
print("%-72s %6s %9s %9s %9s %10s" % ("kernel", "calls", "mean us", "min us", "max us", "total us"))
for name, n, mean, mn, mx, tot in c.execute(q):
    if flt in name:
        print("%-72s %6d %9.2f %9.2f %9.2f %10.1f" % (name[:72], n, mean, mn, mx, tot))
