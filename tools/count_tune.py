#!/usr/bin/env python3
"""A/B of the count(*) kernels in ONE process (same columns, min of 8 passes each).
Usage: python tools/count_tune.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 29
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
torch.cuda.synchronize()
for shape in (256, 128, 64):
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, shape)
    by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, shape)
    grid = sa.Grid([bx, by])
    a = sa.AggCount_int64(grid, 1, 1)
    bx.set_data(0, x); by.set_data(0, y); bx.clear_data_mask(0); by.clear_data_mask(0)
    a.clear_data_mask(0)
    for fast in (1, 0):
        for block in (0, 256, 512, 1024):
            for blocks in (0, 256):
                sa.config_set("count_fast", fast); sa.config_set("block", block); sa.config_set("blocks", blocks)
                best = 1e9
                for _ in range(8):
                    a.reset()
                    sa.timer_start(0)
                    grid.bin(0, [a], rows)
                    best = min(best, sa.timer_stop(0))
                print(f"shape={shape} count_fast={fast} block={block:4d} blocks={blocks:4d}  {best:.3f} ms = {rows/best/1e6:6.1f} Grows/s {rows*16/best/1e6:6.0f} GB/s  {sa.last_kernel(0)}", flush=True)
