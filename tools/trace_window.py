#!/usr/bin/env python3
"""Print a window of a rocprofv3 kernel trace as a timeline: start/end (us, relative), queue, kernel.
    python tools/trace_window.py trace.csv [first_kernel_index] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in rows))
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(ev) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
t0 = ev[i0][0]
for s, e, q, name in ev[i0:i0 + n]:
    short = name.split("(")[0].replace("void ", "").replace("rtk::", "")[:34]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}  {short}")
