#!/usr/bin/env python3
"""Dense-key groupby end to end through Frame (1e6 int64 keys, sum/mean/std), a few runs — for rocprofv3 / timing.
Usage: python tools/r03_gb_dense.py [rows] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vaex_amd
from vaex_amd.binned import Frame, agg
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
torch.cuda.synchronize()
df = Frame(dict(v=v, k=k))
spec = {"s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
for _ in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = df.groupby("k", spec)
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(f"dense groupby {rows} rows: {t*1e3:.3f} ms = {rows/t/1e9:.1f} Grows/s  {sa.last_kernel(0)}  groups {len(out['k'])}", flush=True)
