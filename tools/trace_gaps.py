#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: per-kernel count/avg, total busy time, idle gaps.
    python tools/trace_gaps.py path/to/*_kernel_trace.csv [skip_first_n]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))[skip:]
busy = 0
gaps = collections.Counter()
gapn = collections.Counter()
per = collections.defaultdict(list)
last_end = ev[0][0]
for s, e, n in ev:
    short = n.split("(")[0].replace("void ", "").replace("rtk::", "")[:40]
    per[short].append(e - s)
    if s > last_end:
        gaps[short] += s - last_end
        gapn[short] += 1
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
span = last_end - ev[0][0]
print(f"span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), kernels {len(ev)}")
print("kernel                                    n     avg_us   total_ms   gap_before_avg_us  gap_total_ms")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:40s} {len(v):5d} {sum(v) / len(v) / 1e3:9.1f} {sum(v) / 1e6:9.2f} "
          f"{(gaps[k] / max(1, gapn[k])) / 1e3:12.1f} {gaps[k] / 1e6:12.2f}")
