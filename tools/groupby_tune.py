#!/usr/bin/env python3
"""Kernel time of the groupby aggregation pass (BASELINE configs[3], dense int64 keys -> ordinal binner; 1e6 groups;
sum / count / sum-of-squares of a float64 column) under a few partition knobs, in ONE process.
Usage: python tools/groupby_tune.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 29
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
torch.cuda.synchronize()
b = sa.BinnerOrdinal_int64(1, "k", 1_000_000, 0, False, False)
b.set_data(0, k); b.clear_data_mask(0)
grid = sa.Grid([b])
aggs = [sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1), sa.AggSumMoment_float64(grid, 1, 1, 2)]
for a in aggs:
    a.set_data(0, v, 0); a.clear_data_mask(0)


def run(**cfg):
    for key, val in cfg.items():
        sa.config_set(key, val)
    best = 1e9
    for _ in range(4):
        for a in aggs:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, aggs, rows)
        best = min(best, sa.timer_stop(0))
    tot = int(aggs[1].get_result().sum())
    print(f"{str(cfg):<50} {best:8.3f} ms = {rows/best/1e6:6.1f} Grows/s {rows*16/best/1e6:6.0f} GB/s  {sa.last_kernel(0)} (count {tot})", flush=True)
    for key in cfg:
        sa.config_set(key, {'blk': 1}.get(key, 0))


run()
run(part_lds=160000)
run(part_lds=160000, part_rows=4)
run(part_lds=160000, blk=2)
run(part_lds=160000, scatter_wgs=1)
run(part_lds=160000, scatter_wgs=3)
run(part_lds=160000, part_chunk=1 << 27)
run(part_lds=160000, part_chunk=1 << 29)
run(blk=2)
run()
