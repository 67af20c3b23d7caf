"""Summarise a rocprofv3 --kernel-trace rocpd database (gpurun_out/.../*_results.db) into a per-kernel table."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace summary of %s" % sys.argv[1])
print("# total kernel time %.1f us over %g frames = %.1f us/frame" % (tot, frames, tot / frames))
print("%-100s %7s %11s %9s %9s %9s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%-100s %7d %11.0f %9.1f %9.1f %9.1f %5.1f%%" % (r[0][:100], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
