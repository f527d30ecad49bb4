#!/usr/bin/env python3
"""A/B of partition-pair launch knobs on the BASELINE pass in ONE process (min of 6 passes each).
Usage: python tools/part_tune.py [rows]"""
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


def run(**cfg):
    for k, val in cfg.items():
        sa.config_set(k, val)
    best = 1e9
    for _ in range(6):
        for a in aggs:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, aggs, rows)
        best = min(best, sa.timer_stop(0))
    hot = f"hot {sa.config_get('hot_w')}x{sa.config_get('hot_h')}@({sa.config_get('hot_x0')},{sa.config_get('hot_y0')}) {sa.config_get('hot_fraction_ppm')/1e4:.1f}%"
    print(f"{str(cfg):<60} {best:.3f} ms = {rows/best/1e6:6.1f} Grows/s  [{hot}]", flush=True)
    for k in cfg:
        sa.config_set(k, {'hot': 1, 'blk': 1}.get(k, 0))


run()
run(hot=0)
run(hot=0, blk=0)
run(part_chunk=1 << 29)
