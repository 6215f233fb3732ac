#!/usr/bin/env python3
"""Summarise rocprofv3 sqlite output: per-kernel avg duration and counter means.  usage: pmc_query.py db [substr]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
sub = sys.argv[2] if len(sys.argv) > 2 else ""
try:
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
except Exception:
    rows = []
for r in rows:
    if sub in r[0]:
        print(f"{r[0][:48]:48s} {r[1]:28s} {r[2]:16.1f} n={r[3]}")
if not rows:
    for r in cur.execute("select name, total_calls, average, percentage from top_kernels"):
        if sub in r[0]:
            print(f"{r[0][:70]:70s} calls={r[1]:5d} avg_us={r[2]:10.2f} pct={r[3]:5.1f}")
