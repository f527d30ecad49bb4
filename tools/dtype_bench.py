#!/usr/bin/env python3
"""The bench pass (2-D 256^2 count+sum+count) and a 2-D count on columns of other dtypes. Usage: python tools/dtype_bench.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
g = torch.Generator(device="cuda").manual_seed(1)
x64 = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y64 = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v64 = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
NAMES = {torch.float64: "float64", torch.float32: "float32", torch.int32: "int32", torch.int64: "int64"}


def run(label, dt, shape, with_sum):
    if dt.is_floating_point:
        x, y, v = x64.to(dt), y64.to(dt), v64.to(dt)
        lo, hi = -4.0, 4.0
    else:
        x, y, v = (x64 * 1000).to(dt), (y64 * 1000).to(dt), (v64 * 100).to(dt)
        lo, hi = -4000.0, 4000.0
    pf = NAMES[dt]
    bx = getattr(sa, "BinnerScalar_" + pf)(1, "x", lo, hi, shape)
    by = getattr(sa, "BinnerScalar_" + pf)(1, "y", lo, hi, shape)
    for b, c in ((bx, x), (by, y)):
        b.set_data(0, c); b.clear_data_mask(0)
    grid = sa.Grid([bx, by])
    aggs = [sa.AggCount_int64(grid, 1, 1)]
    if with_sum:
        a = getattr(sa, "AggSum_" + pf)(grid, 1, 1); a.set_data(0, v, 0); aggs.append(a)
        a = getattr(sa, "AggCount_" + pf)(grid, 1, 1); a.set_data(0, v, 0); aggs.append(a)
    for a in aggs:
        a.clear_data_mask(0)
    best = 1e9
    for _ in range(4):
        for a in aggs:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, aggs, rows)
        best = min(best, sa.timer_stop(0))
    bpr = (3 if with_sum else 2) * x.element_size()
    print(f"{label:<44} {best:8.3f} ms {rows/best/1e6:8.1f} Grows/s {rows*bpr/best/1e6:7.0f} GB/s  {sa.last_kernel(0)}", flush=True)


for dt in (torch.float64, torch.float32, torch.int32):
    run(f"{NAMES[dt]} count+sum+count 256^2", dt, 256, True)
    run(f"{NAMES[dt]} count 256^2", dt, 256, False)
    run(f"{NAMES[dt]} count 128^2", dt, 128, False)
    run(f"{NAMES[dt]} count+sum+count 64^2 (LDS)", dt, 64, True)
