#!/usr/bin/env python3
"""Where pass 1 of the hot path spends its time: the pass with the cold rows dropped, the box updates dropped, every record sent
to the sink record (timing experiments behind bits of the no_pipeline knob: results are wrong by construction).
Usage (GPU box): python tools/ablate_hot.py [rows] [wv: 1 part_scatter_blk, 3 part_scatter_wv DIRECT=1, 4 DIRECT=2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vaex_amd
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
grid = sa.Grid([bx, by])
al = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
al[1].set_data(0, v, 0); al[2].set_data(0, v, 0); bx.set_data(0, x); by.set_data(0, y)
wv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sa.config_set("wv", wv)
# bits of the no_pipeline knob: 64 drops the cold rows; 128 (part_scatter_blk) the box updates; 2 (shared streams) sends every record to the sink
variants = {1: (("full", 0), ("no cold rows (1/4 of the box updates)", 64), ("no box updates", 128), ("full again", 0)),
            3: (("full", 0), ("no cold rows", 64), ("full again", 0)),
            4: (("full", 0), ("all records to the sink", 2), ("no cold rows", 64), ("full again", 0))}[wv]
for label, bits in variants:
    sa.config_set("no_pipeline", bits)
    best = 1e9
    for _ in range(4):
        for a in al: a.reset()
        sa.timer_start(0); grid.bin(0, al, rows); best = min(best, sa.timer_stop(0))
    print(f"{label:<40} {best:7.3f} ms  {rows/best/1e6:6.1f} Grows/s  {sa.last_kernel(0)}  counted {int(al[0].get_result().sum())}", flush=True)
sa.config_set("no_pipeline", 0)
