#!/usr/bin/env python3
"""Times the other BASELINE.json configs on one GPU (they are parity-test cases, not the bench line):
  configs[2]  3-D histogram (x,y,z) shape 128^3 with a boolean selection mask
  configs[3]  groupby on a 1e6-cardinality int64 key, agg sum/mean/std of v (dense keys -> ordinal binner,
              scattered keys -> hash binner)
Usage: python tools/configs_bench.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vaex_amd
from vaex_amd.binned import Frame, agg

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 29
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
z = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
sel = v > 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
ks = (k * 2654435761) % (1 << 40)
torch.cuda.synchronize()
df = Frame(dict(x=x, y=y, z=z, v=v, sel=sel, k=k, ks=ks))


def timed(label, fn, bytes_per_row, reps=3):
    best = 1e9
    out = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{label:<58} {best*1e3:9.2f} ms {rows/best/1e9:8.2f} Grows/s {rows*bytes_per_row/best/1e9:8.0f} GB/s  [{sa.last_kernel(0)}]", flush=True)
    return out


c3 = timed("C3 3-D 128^3 count, selection v>3 (25 B/row)", lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128, selection="sel"), 25)
assert int(c3.sum()) <= rows
c3n = timed("C3' 3-D 128^3 count, no selection (24 B/row)", lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128), 24)
spec = {"s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
g1 = timed("C4 groupby dense 1e6 int64 keys: sum/mean/std (16 B/row)", lambda: df.groupby("k", spec), 16, reps=2)
g2 = timed("C4' groupby scattered 1e6 int64 keys (hash binner)", lambda: df.groupby("ks", spec), 16, reps=2)
assert len(g1["k"]) == len(g2["ks"])
np.testing.assert_allclose(np.sort(g1["s"]), np.sort(g2["s"]), rtol=1e-9)
c2 = timed("C2 2-D 256^2 count+mean (24 B/row, incl. host finish)", lambda: df._agg([agg.count(), agg.mean("v")], binby=["x", "y"], limits=[[-4, 4]] * 2, shape=256), 24)
