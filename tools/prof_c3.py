#!/usr/bin/env python3
"""3-D 128^3 count with a selection (BASELINE configs[2]) a few times — for rocprofv3. Usage: python tools/prof_c3.py [rows] [key=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 28
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    sa.config_set(k, int(v))
g = torch.Generator(device="cuda").manual_seed(1)
cols = [torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
sel = (torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) > 0).to(torch.uint8)
torch.cuda.synchronize()
binners = [sa.BinnerScalar_float64(1, "xyz"[d], -4.0, 4.0, 128) for d in range(3)]
for b, c in zip(binners, cols):
    b.set_data(0, c); b.clear_data_mask(0)
grid = sa.Grid(binners)
a = sa.AggCount_int64(grid, 1, 1)
a.set_data_mask(0, sel)
for _ in range(4):
    a.reset()
    sa.timer_start(0)
    grid.bin(0, [a], rows)
    ms = sa.timer_stop(0)
print(f"rows={rows} last pass {ms:.3f} ms = {rows/ms/1e6:.1f} Grows/s {sa.last_kernel(0)} sum={int(a.get_result().sum())}")
