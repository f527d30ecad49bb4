#!/usr/bin/env python3
"""Kernel-time (vxh_timer) of the fused pass for a few grid shapes / aggregator sets on HBM-resident float64 columns.
Usage: python tools/shapes_bench.py [rows] [key=value ...]"""
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
cols = {c: torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for c in "xyz"}
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
sel = (v > 3).to(torch.uint8)
torch.cuda.synchronize()


def run(label, dims, shape, kinds, bytes_per_row, mask=False, cfg=()):
    for k, val in cfg:
        sa.config_set(k, val)
    binners = [sa.BinnerScalar_float64(1, d, -4.0, 4.0, shape) for d in dims]
    for b, d in zip(binners, dims):
        b.set_data(0, cols[d]); b.clear_data_mask(0)
    grid = sa.Grid(binners)
    aggs = []
    for kind in kinds:
        if kind == "count":
            a = sa.AggCount_int64(grid, 1, 1)
        elif kind == "countv":
            a = sa.AggCount_float64(grid, 1, 1); a.set_data(0, v, 0)
        elif kind == "sum":
            a = sa.AggSum_float64(grid, 1, 1); a.set_data(0, v, 0)
        else:
            a = sa.AggSumMoment_float64(grid, 1, 1, 2); a.set_data(0, v, 0)
        if mask:
            a.set_data_mask(0, sel)
        else:
            a.clear_data_mask(0)
        aggs.append(a)
    best = 1e9
    for _ in range(4):
        for a in aggs:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, aggs, rows)
        best = min(best, sa.timer_stop(0))
    total = int(aggs[0].get_result().sum())
    print(f"{label:<66} {best:8.3f} ms {rows/best/1e6:8.1f} Grows/s {rows*bytes_per_row/best/1e6:8.0f} GB/s  {sa.last_kernel(0)}  (sum {total})", flush=True)
    for k, _ in cfg:
        sa.config_set(k, {"count16": 1, "blk": 1, "hot": 1}.get(k, 0))


run("count only 256^2 (packed u16 LDS counters)", "xy", 256, ["count"], 16)
run("count only 256^2, count16=0 (partition + reduce)", "xy", 256, ["count"], 16, cfg=[("count16", 0)])
run("count(sel) 256^2 (17 B/row)", "xy", 256, ["count"], 17, mask=True)
run("count only 128^2 (LDS u32)", "xy", 128, ["count"], 16)
run("count+sum 256^2", "xy", 256, ["count", "sum"], 24)
run("count+sum+countv 256^2 (the bench pass)", "xy", 256, ["count", "sum", "countv"], 24)
run("count+sum+countv 64^2 (LDS)", "xy", 64, ["count", "sum", "countv"], 24)
run("count+sum+countv 1024^2", "xy", 1024, ["count", "sum", "countv"], 24)
run("count 3-D 128^3 selection (25 B/row)", "xyz", 128, ["count"], 25, mask=True)
run("count 3-D 128^3 selection, blk=0", "xyz", 128, ["count"], 25, mask=True, cfg=[("blk", 0)])
run("count 3-D 64^3 (24 B/row)", "xyz", 64, ["count"], 24)
run("count 3-D 64^3, blk=0", "xyz", 64, ["count"], 24, cfg=[("blk", 0)])
run("count+sum+countv 1024^2, blk=0", "xy", 1024, ["count", "sum", "countv"], 24, cfg=[("blk", 0)])
run("count only 512^2", "xy", 512, ["count"], 16)
run("count only 512^2, hot=0", "xy", 512, ["count"], 16, cfg=[("hot", 0)])
run("count only 512^2, blk=0", "xy", 512, ["count"], 16, cfg=[("blk", 0)])
