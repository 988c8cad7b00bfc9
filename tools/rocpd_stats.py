"""Summarise a rocprofv3 (rocpd sqlite) kernel trace like `--stats`: per kernel
calls, total/avg/min/max duration.  Usage: python tools/rocpd_stats.py results.db [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(
    "select {n}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
    "from kernels group by {n} order by sum(end-start) desc".format(n=name_col)
).fetchall()
total = sum(r[2] for r in rows) or 1
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for name, calls, tot, avg, mn, mx in rows:
    lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, calls, tot, avg, 100.0 * tot / total, mn, mx))
text = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
print(text)
