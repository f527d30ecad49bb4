#!/usr/bin/env python3
"""groupby (BASELINE configs[3]) + host-streamed pass, for rocprofv3 kernel stats. Usage: python tools/prof_groupby.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
from vaex_amd.binned import Frame, agg
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
ks = (k * 2654435761) % (1 << 40)
torch.cuda.synchronize()
df = Frame(dict(v=v, k=k, ks=ks))
spec = {"s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
for name, key in (("dense", "k"), ("hash", "ks")):
    for _ in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = df.groupby(key, spec)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"groupby {name}: {dt*1e3:.2f} ms {rows/dt/1e9:.2f} Grows/s groups={len(r[key])} [{sa.last_kernel(0)}]")
# host-streamed (PCIe-inclusive) 2-D count+mean: numpy columns, 1 Mi-row chunks over 4 slots
n = 1 << 26
rng = np.random.default_rng(1)
hx, hy, hv = rng.normal(0, 1, n), rng.normal(0, 1, n), rng.normal(3, 2, n)
for nthreads in (1, 4, 8):
    hf = Frame(dict(x=hx, y=hy, v=hv), chunk_size=1 << 20, nthreads=nthreads)
    for _ in range(2):
        t0 = time.perf_counter()
        c, m = hf._agg([agg.count(), agg.mean("v")], binby=["x", "y"], limits=[[-4, 4]] * 2, shape=256)
        dt = time.perf_counter() - t0
    assert c.sum() <= n
    print(f"host-streamed 2-D 256^2 count+mean, {nthreads} slots: {dt*1e3:.1f} ms {n/dt/1e9:.3f} Grows/s = {n*24/dt/1e9:.1f} GB/s over PCIe [{sa.last_kernel(0)}]")
