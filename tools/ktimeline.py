#!/usr/bin/env python3
"""Timeline (start, duration, gap to the previous kernel's end) of the last N kernels of a rocprofv3
--kernel-trace CSV.  Usage: python tools/ktimeline.py <kernel_trace.csv> [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 40):]
t0 = int(rows[0]["Start_Timestamp"])
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {name}")
    prev = e
