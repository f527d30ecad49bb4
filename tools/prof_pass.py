#!/usr/bin/env python3
"""Run the BASELINE pass (2-D 256x256 count+sum+count) a few times — for rocprofv3.
Usage: python tools/prof_pass.py [rows] [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 29
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    sa.config_set(k, int(v))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
torch.cuda.synchronize()
bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256)
by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
grid = sa.Grid([bx, by])
aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
bx.set_data(0, x); by.set_data(0, y); bx.clear_data_mask(0); by.clear_data_mask(0)
aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
for a in aggs:
    a.clear_data_mask(0)
for _ in range(4):
    for a in aggs:
        a.reset()
    sa.timer_start(0)
    grid.bin(0, aggs, rows)
    ms = sa.timer_stop(0)
print(f"rows={rows} last pass {ms:.3f} ms = {rows/ms/1e6:.1f} Grows/s {sa.last_kernel(0)}")
