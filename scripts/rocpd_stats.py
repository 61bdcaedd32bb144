#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace: per-kernel calls / total / avg / min / max / share.
usage: python scripts/rocpd_stats.py <results.db> [out.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                        f"from kernels group by {name_col} order by 3 desc"))
total = sum(r[2] for r in rows)
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
for n, c, t, a, mn, mx in rows:
    lines.append(f'"{n}",{c},{t},{a:.0f},{mn},{mx},{100.0 * t / total:.2f}')
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)
