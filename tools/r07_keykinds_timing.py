#!/usr/bin/env python3
"""Round 6 (late): df.groupby of the REAL vaex under install() over the key kinds the device groupby learnt this round — wall time of the wrapped call
(device groupby) next to vaex's own two passes on the HIP classes (the wrapper's decline road: `__wrapped__`).  Usage: python tools/r07_keykinds_timing.py [rows]"""
import os, sys, time, warnings
warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "_ref", "vaexpy"), os.path.join(ROOT, "oracle", "fake"), ROOT]
import numpy as np
import vaex, vaex_amd
from vaex_amd import vaex_groupby as vg
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
vaex_amd.install()
original = vaex.dataframe.DataFrameLocal.groupby.__wrapped__
A = vaex.agg
rng = np.random.default_rng(3)
k = rng.integers(0, 1000, n)
v = rng.normal(3, 2, n)
mask = rng.random(n) < 0.1
df = vaex.from_arrays(k=k, c=k.copy(), km=np.ma.array(k, mask=mask), kf=k * 0.5, v=v, vm=np.ma.array(v, mask=mask))
df.categorize("c", min_value=0, max_value=999, inplace=True)
agg = {"n": A.count(), "s": A.sum("v"), "m": A.mean("v"), "sd": A.std("v")}
cases = [("int64 key", "k", agg), ("categorical key (1000 categories)", "c", agg), ("int64 key with missing values (numpy mask)", "km", agg), ("float64 key", "kf", agg),
         ("int64 key, values with missing entries", "k", {"n": A.count("vm"), "s": A.sum("vm"), "m": A.mean("vm")}), ("int64 key, nunique of an int64 column", "k", {"u": A.nunique("c")})]
print(f"{n:.0e} rows, host numpy columns; wall seconds, best of 2")
for label, by, a in cases:
    best = {}
    for which, fn in (("device", lambda: df.groupby(by, agg=a, sort=True)), ("vaex on the HIP classes", lambda: original(df, by, agg=a, sort=True))):
        ts = []
        for _ in range(2):
            vg.last.clear()
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
            if which == "device":
                assert vg.last.get("path") == "device", (label, vg.last)
        best[which] = (min(ts), len(r))
    d, o = best["device"], best["vaex on the HIP classes"]
    print(f"{label:<52} device groupby {d[0]:7.3f} s   vaex's own passes {o[0]:7.3f} s   x{o[0] / d[0]:5.1f}   groups {d[1]} / {o[1]}", flush=True)
