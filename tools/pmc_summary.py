#!/usr/bin/env python3
"""Fold rocprofv3 counter_collection CSVs into one small table: per kernel, per counter, mean per dispatch.
    python tools/pmc_summary.py out.csv in1_counter_collection.csv [in2 ...]
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; hbm_*_MB columns apply the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE x2 for wide coalesced reads, WRITE_SIZE x1)."""
import collections
import csv
import sys

csv.field_size_limit(1 << 30)
acc = collections.defaultdict(lambda: [0.0, 0])
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "rtk::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("rtk::", "")
        k = (short, r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"])
        acc[k][1] += 1
kernels = sorted({k[0] for k in acc})
counters = sorted({k[1] for k in acc})
with open(sys.argv[1], "w", newline="") as f:
    w = csv.writer(f)
    extra = ["hbm_read_MB(x2)", "hbm_write_MB", "hbm_total_MB"] if "FETCH_SIZE" in counters else []
    w.writerow(["kernel", "dispatches"] + counters + extra)
    for kn in kernels:
        n = max(acc[(kn, c)][1] for c in counters if (kn, c) in acc)
        vals = [acc[(kn, c)][0] / acc[(kn, c)][1] if (kn, c) in acc else "" for c in counters]
        row = [kn, n] + [f"{v:.1f}" if v != "" else "" for v in vals]
        if extra:
            fs = acc[(kn, "FETCH_SIZE")][0] / max(1, acc[(kn, "FETCH_SIZE")][1]) if (kn, "FETCH_SIZE") in acc else 0.0
            ws = acc[(kn, "WRITE_SIZE")][0] / max(1, acc[(kn, "WRITE_SIZE")][1]) if (kn, "WRITE_SIZE") in acc else 0.0
            rd, wr = 2.0 * fs * 1024 / 1e6, ws * 1024 / 1e6
            row += [f"{rd:.2f}", f"{wr:.2f}", f"{rd + wr:.2f}"]
        w.writerow(row)
print("wrote", sys.argv[1], len(kernels), "kernels")
