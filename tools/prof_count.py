#!/usr/bin/env python3
"""Count-only 2-D passes (256^2: packed uint16 LDS counters; 128^2: uint32 LDS counters) — for rocprofv3.
Usage: python tools/prof_count.py [rows] [key=value ...]"""
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
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
torch.cuda.synchronize()
for shape in (256, 128):
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, shape)
    by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, shape)
    grid = sa.Grid([bx, by])
    a = sa.AggCount_int64(grid, 1, 1)
    bx.set_data(0, x); by.set_data(0, y); bx.clear_data_mask(0); by.clear_data_mask(0)
    a.clear_data_mask(0)
    for _ in range(3):
        a.reset()
        sa.timer_start(0)
        grid.bin(0, [a], rows)
        ms = sa.timer_stop(0)
    print(f"shape={shape} rows={rows} last pass {ms:.3f} ms = {rows/ms/1e6:.1f} Grows/s {sa.last_kernel(0)}")
