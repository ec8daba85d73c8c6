#!/usr/bin/env python3
"""Dump the kernel statistics of a rocprofv3 rocpd (.db) result as a small CSV (for profiles/)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("kernel,calls,total_us,avg_us,percent")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
    print(f'"{name}",{calls},{total:.3f},{avg:.3f},{pct:.3f}')
try:
    rows = cur.execute("select name, value from counters_collection").fetchall()
except Exception:
    rows = []
if rows:
    print("\ncounter,sum_over_dispatches")
    agg = {}
    for n, v in rows:
        agg[n] = agg.get(n, 0) + (v or 0)
    for n, v in agg.items():
        print(f"{n},{v}")
