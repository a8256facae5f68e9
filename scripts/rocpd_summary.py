#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (trace_results.db) into a kernel-stats CSV (name, calls, avg/min/max ns, %)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPRs,SGPRs,LDS,GridX,WorkgroupX")
for r in rows:
    print(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.1f},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}")
