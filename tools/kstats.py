#!/usr/bin/env python3
"""Print a trimmed per-kernel summary of a rocprofv3 *_kernel_stats.csv (long template names cut)."""
import csv
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = list(csv.DictReader(open(path)))
print(f"{'kernel':<60} {'calls':>6} {'avg_us':>10} {'total_ms':>10} {'pct':>6}")
for r in rows[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:60]:<60} {r['Calls']:>6} {float(r['AverageNs'])/1e3:>10.1f} {float(r['TotalDurationNs'])/1e6:>10.2f} {float(r['Percentage']):>6.2f}")
