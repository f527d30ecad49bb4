#!/usr/bin/env python3
"""Diagnosis (round 6): vaex's OWN groupby over several keys, one of them masked, on the HIP classes (install()) against the reference's C++ —
the random-call grammar's seed 615.  Usage: python tools/r07_masked_combined.py"""
import os, sys, warnings
warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "_ref", "vaexpy"), os.path.join(ROOT, "oracle", "fake"), ROOT]
import numpy as np
import vaex, vaex_amd
vaex_amd.install()
original = vaex.dataframe.DataFrameLocal.groupby.__wrapped__
A = vaex.agg
rng = np.random.default_rng(5)
n = 5000
k0 = rng.choice(rng.integers(-2**30, 2**30, 150), n).astype("i4")
k1 = np.ma.array(np.full(n, 7, dtype="i4"), mask=rng.random(n) < 0.15)
k2 = (rng.integers(0, 12, n) * 9).astype("i8")
vf = rng.normal(0, 1, n).astype("f4")
df = vaex.from_arrays(k0=k0, k1=k1, k2=k2, vf=vf, v=rng.normal(0, 1, n))
def table(t, keys):
    rows = list(zip(*[t[c].tolist() for c in keys + ["c"]]))
    return sorted(rows, key=lambda r: tuple((x is None, 0 if x is None else x) for x in r))
def run(keys, filtered, delayed, installed):
    d = df[df.vf > -0.5] if filtered else df
    if installed:
        if delayed:
            p = d.groupby(keys, agg={"c": A.count()}, delay=True); d.execute(); return table(p.get(), keys)
        return table(d.groupby(keys, agg={"c": A.count()}), keys)
    vaex_amd.uninstall()
    try:
        if delayed:
            p = d.groupby(keys, agg={"c": A.count()}, delay=True); d.execute(); return table(p.get(), keys)
        return table(d.groupby(keys, agg={"c": A.count()}), keys)
    finally:
        vaex_amd.install()
from vaex_amd import vaex_groupby as vg
for keys in (["k1"], ["k0", "k1"], ["k1", "k2"], ["k0", "k1", "k2"]):
    for filtered in (False, True):
        for delayed in (False, True):
            vg.last.clear()
            got = run(keys, filtered, delayed, True)
            path = vg.last.get("path")
            want = run(keys, filtered, delayed, False)
            same = got == want
            print(keys, "filtered" if filtered else "", "delayed" if delayed else "", "path", path, "groups", len(got), len(want), "rows", sum(r[-1] for r in got), sum(r[-1] for r in want), "SAME" if same else "DIFFERENT")
            if not same:
                g, w = set(got), set(want)
                print("   only here", sorted(g - w, key=str)[:6], "\n   only there", sorted(w - g, key=str)[:6])
