#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, sum of each counter / dispatches."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for path in sys.argv[1:]:
    for f in glob.glob(path):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "")[:40]
            if not any(k in name for k in ("part_", "bin_kernel", "count_lds", "fold", "hm_", "gb_")):
                continue
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[(name, r["Counter_Name"])].add(r["Dispatch_Id"])
for name, ctrs in acc.items():
    print(name)
    for c, v in sorted(ctrs.items()):
        n = len(disp[(name, c)])
        print(f"   {c:<28} {v/n:18.1f} per dispatch  ({n} dispatches)")
