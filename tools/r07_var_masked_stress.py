#!/usr/bin/env python3
"""Diagnosis (round 6): the one intermittent difference of the random-call tests — var of a MASKED value column with a selection on a filtered frame
(seeds 50 / 384, once per ~2 full-suite runs).  The same call repeated under install(): does its result vary from repeat to repeat?
Usage: python tools/r07_var_masked_stress.py [repeats]"""
import os, sys, warnings
warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle", "_ref", "vaexpy"), os.path.join(ROOT, "oracle", "fake"), ROOT]
import numpy as np
import vaex, vaex_amd
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = 120_000
r = np.random.default_rng(21)
x = r.normal(0, 1, n); x[::997] = np.nan
v = r.normal(3, 2, n); v[::501] = np.nan
df = vaex.from_arrays(x=x, y=r.normal(0, 1, n), v=v, h=r.integers(-300, 300, n).astype("i2"), m=np.ma.array(r.normal(0, 1, n), mask=r.random(n) < 0.05), w=r.normal(0, 1, n))
frames = {"plain": df, "filtered": df[df.x > -0.5], "filtered_libm": df[np.sin(df.y * 3) > -0.5]}
sel = "((((x <= 9007199254740993) | (y/y == -3)) | ((-y >= -2.84) & (3 < h))) | (v**2 <= -0.197))"
calls = {
    "var m sel": lambda d: d.var("m", binby="v", limits=[-3, 9], shape=67, selection=sel),
    "var m": lambda d: d.var("m", binby="v", limits=[-3, 9], shape=67),
    "sum m sel": lambda d: d.sum("m", binby="v", limits=[-3, 9], shape=67, selection=sel),
    "count m sel": lambda d: d.count("m", binby="v", limits=[-3, 9], shape=67, selection=sel),
    "var w sel": lambda d: d.var("w", binby="v", limits=[-3, 9], shape=67, selection=sel),
    "mean m sel simple": lambda d: d.mean("m", binby="v", limits=[-3, 9], shape=67, selection="y > 0"),
}
want = {}
for fname, d in frames.items():
    for cname, fn in calls.items():
        want[fname, cname] = np.asarray(fn(d))          # vaex's own C++ (not installed yet)
vaex_amd.install()
for fname, d in frames.items():
    for cname, fn in calls.items():
        ref = want[fname, cname]
        worst, differing = 0.0, 0
        for i in range(reps):
            got = np.asarray(fn(d))
            with np.errstate(invalid="ignore"):
                dev = float(np.nanmax(np.abs(got - ref))) if np.isfinite(ref).any() else 0.0
            if not np.allclose(got, ref, rtol=1e-9, atol=1e-9, equal_nan=True):
                differing += 1
                worst = max(worst, dev)
        print(f"{fname:<14} {cname:<20} repeats {reps}  differing from the reference {differing}  worst |diff| {worst:.3g}", flush=True)
print("task parts:", {k: v for k, v in vaex_amd.task_stats.items() if isinstance(v, (int, float))})
