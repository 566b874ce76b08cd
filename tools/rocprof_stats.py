"""Per-kernel summary (calls, total, average, min, max in ns) of a rocprofv3 results database, as CSV on stdout.
Usage: python tools/rocprof_stats.py results.db [name-prefix-filter]"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc")
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
for r in rows:
    if flt in r[0]:
        w.writerow(r)
