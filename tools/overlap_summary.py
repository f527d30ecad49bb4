#!/usr/bin/env python3
"""Copy/compute overlap from a rocprofv3 --kernel-trace --memory-copy-trace run (csv output):
busy time of the H2D copies, of the kernels, of their union and their intersection, plus a coarse text timeline.

usage: overlap_summary.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [timeline buckets] [timeline ms]"""
import csv
import glob
import os
import sys


def load(pattern, want=None):
    out = []
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if want and not want(r):
                continue
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r))
    return sorted(out, key=lambda t: t[0])


def union(iv):
    merged = []
    for s, e in sorted(iv):
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    return merged


def total(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    out = []
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            out.append([s, e])
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return out


d = sys.argv[1]
buckets = int(sys.argv[2]) if len(sys.argv) > 2 else 150
window_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
kern = load(os.path.join(d, "**", "*kernel_trace.csv"))
cop = load(os.path.join(d, "**", "*memory_copy_trace.csv"), lambda r: "HOST_TO_DEVICE" in r.get("Direction", "").upper() or "H2D" in r.get("Direction", "").upper())
if not kern or not cop:
    raise SystemExit(f"kernels: {len(kern)}, H2D copies: {len(cop)} — nothing to compare")
ku = union([(s, e) for s, e, _ in kern])
cu = union([(s, e) for s, e, _ in cop])
both = intersect(ku, cu)
any_ = union([(s, e) for s, e in ku] + [(s, e) for s, e in cu])
t0, t1 = min(ku[0][0], cu[0][0]), max(ku[-1][1], cu[-1][1])
nbytes = 0
for _, _, r in cop:
    for key in ("Bytes", "Size", "bytes"):
        if key in r and r[key]:
            nbytes += int(float(r[key]))
            break
print(f"span {(t1-t0)/1e6:.1f} ms: {len(kern)} kernels, {len(cop)} H2D copies" + (f" ({nbytes/1e9:.2f} GB)" if nbytes else ""))
print(f"  copy engine busy   {total(cu)/1e6:9.1f} ms  ({100*total(cu)/(t1-t0):5.1f} % of the span)")
print(f"  kernels busy       {total(ku)/1e6:9.1f} ms  ({100*total(ku)/(t1-t0):5.1f} %)")
print(f"  both at once       {total(both)/1e6:9.1f} ms  ({100*total(both)/max(1,total(ku)):5.1f} % of the kernel time runs under a copy)")
print(f"  either             {total(any_)/1e6:9.1f} ms")
if nbytes:
    print(f"  H2D rate while copying {nbytes/total(cu):.1f} GB/s; over the span {nbytes/(t1-t0):.1f} GB/s")
names = {}
for s, e, r in kern:
    k = r.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    a = names.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
for k, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"    {k:<48} {c:6d} x {t/c/1e3:8.1f} us")
# timeline of a window in the middle of the run: one character per bucket, C copy only, K kernel only, # both, . idle
mid = cop[len(cop) // 2][0]  # (the middle copy: inside a pass, not between two)
w0 = mid
w1 = w0 + min(t1 - mid, window_ms * 1e6)
step = (w1 - w0) / buckets


def cover(iv, a, b):
    return sum(max(0, min(e, b) - max(s, a)) for s, e in iv if e > a and s < b)


row = ""
for i in range(buckets):
    a, b = w0 + i * step, w0 + (i + 1) * step
    c, k = cover(cu, a, b) > 0.5 * step, cover(ku, a, b) > 0.25 * step
    row += "#" if c and k else "C" if c else "K" if k else "."
print(f"  timeline of {(w1-w0)/1e6:.1f} ms from the middle ({step/1e3:.0f} us per character; C copy, K kernel, # both, . idle):")
print("  " + row)
