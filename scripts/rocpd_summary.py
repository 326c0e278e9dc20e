#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) as CSV: per-kernel calls / total / average / min / max (ns)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), max(lds_size) "
                  "from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,SGPR,LDS")
for r in rows:
    print('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8]))
