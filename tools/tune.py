#!/usr/bin/env python3
"""Sweep the tuning knobs of the partition strategy on the BASELINE workload (2-D 256x256 count+sum+count).
Usage: python tools/tune.py [rows]   (GPU box)"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 29
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
torch.cuda.synchronize()


def run(shape=256, aggs="csc", reps=3, **cfg):
    for k, val in cfg.items():
        sa.config_set(k, val)
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, shape)
    by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, shape)
    grid = sa.Grid([bx, by])
    al = []
    for ch in aggs:
        if ch == "c":
            al.append(sa.AggCount_int64(grid, 1, 1))
        elif ch == "s":
            a = sa.AggSum_float64(grid, 1, 1); a.set_data(0, v, 0); al.append(a)
        elif ch == "n":
            a = sa.AggCount_float64(grid, 1, 1); a.set_data(0, v, 0); al.append(a)
    bx.set_data(0, x); by.set_data(0, y); bx.clear_data_mask(0); by.clear_data_mask(0)
    for a in al:
        a.clear_data_mask(0)
    best = 1e9
    for _ in range(reps):
        for a in al:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, al, rows)
        best = min(best, sa.timer_stop(0))
    pass
    return best, sa.last_kernel(0)


def show(label, ms, kern, bpr=24):
    print(f"{label:<70} {ms:8.3f} ms {rows/ms/1e6:8.1f} Grows/s  {rows*bpr/ms/1e6:7.0f} GB/s  {kern}", flush=True)


defaults = dict(strategy=0, part_rows=0, part_chunk=1 << 26, part_lds=0, parts=0)
for pr, pc, pl in itertools.product([8, 4], [1 << 26, 1 << 27, 1 << 28], [0]):
    cfg = dict(defaults, part_rows=pr, part_chunk=pc, part_lds=pl)
    ms, k = run(aggs="csn", **cfg)
    show(f"count+sum+countv 256^2 part_rows={pr} chunk=2^{pc.bit_length()-1} lds={pl}", ms, k)
for pl in [40 * 1024, 72 * 1024, 150 * 1024]:
    cfg = dict(defaults, part_lds=pl)
    ms, k = run(aggs="csn", **cfg)
    show(f"count+sum+countv 256^2 part_lds={pl}", ms, k)
for k2, v2 in defaults.items():
    sa.config_set(k2, v2)
ms, k = run(aggs="c"); show("count only 256^2", ms, k, 16)
ms, k = run(aggs="cs"); show("count+sum 256^2", ms, k)
ms, k = run(shape=128, aggs="c"); show("count only 128^2 (LDS)", ms, k, 16)
ms, k = run(shape=128, aggs="csn"); show("count+sum+countv 128^2", ms, k)
ms, k = run(shape=64, aggs="csn"); show("count+sum+countv 64^2 (LDS)", ms, k)
ms, k = run(shape=1024, aggs="csn"); show("count+sum+countv 1024^2", ms, k)
