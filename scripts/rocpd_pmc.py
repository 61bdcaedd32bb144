#!/usr/bin/env python
"""Sum a PMC counter per kernel from a rocprofv3 rocpd (sqlite) database.
usage: python scripts/rocpd_pmc.py <results.db> [counter-name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
want = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("no counters_collection view; tables:", tabs)
    sys.exit(1)
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
kcol = [c for c in cols if c in ("kernel_name", "name")][0] if any(c in cols for c in ("kernel_name", "name")) else cols[0]
ccol = [c for c in cols if "counter_name" in c or c == "counter"][0]
vcol = [c for c in cols if c in ("value", "counter_value")][0]
rows = list(cur.execute(f"select {kcol}, {ccol}, count(*), sum({vcol}) from {view} group by {kcol}, {ccol} order by 4 desc"))
print("kernel,counter,dispatches,sum,per_dispatch")
for k, c, n, v in rows:
    if want and want not in c:
        continue
    print(f'"{k}",{c},{n},{v:.0f},{v / n:.0f}')
