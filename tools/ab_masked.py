#!/usr/bin/env python3
"""The bench pass with ONE selection mask shared by every aggregator, and with the sum of squares (std), with and without the hot box.
GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
sa = vaex_amd.superagg
rows = int(1e9)
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
sel = (v > 3).to(torch.uint8)
ref = None
for hot in (1, 0, 1):
    sa.config_set("hot", hot)
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    al = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
    al[1].set_data(0, v, 0); al[2].set_data(0, v, 0); bx.set_data(0, x); by.set_data(0, y)
    for a in al: a.set_data_mask(0, sel)
    best = 1e9
    for _ in range(4):
        for a in al: a.reset()
        sa.timer_start(0); grid.bin(0, al, rows); best = min(best, sa.timer_stop(0))
    res = [np.array(a.get_result()) for a in al]
    if ref is None: ref = res
    ok = np.array_equal(res[0], ref[0]) and np.array_equal(res[2], ref[2]) and bool(np.all(np.abs(res[1] - ref[1]) <= 1e-12 * 20.0 * np.maximum(res[0], 1)))
    print(f"hot={hot} 2-D 256x256 count+sum+count with a selection (25 B/row): {best:7.3f} ms {rows/best/1e6:6.1f} Grows/s {rows*25/best/1e6/8000:5.3f}  {sa.last_kernel(0)} box {sa.config_get('hot_w')}x{sa.config_get('hot_h')} {sa.config_get('hot_fraction_ppm')/1e4:.1f}% {'same' if ok else 'DIFFERENT'} n={int(res[0].sum())}", flush=True)
sa.config_set("hot", 1)
# var / std: count(v) + sum(v) + sum of squares, no selection (24 B/row)
for hot in (1, 0):
    sa.config_set("hot", hot)
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    al = [sa.AggCount_float64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggSumMoment_float64(grid, 1, 1, 2)]
    for a in al: a.set_data(0, v, 0)
    bx.set_data(0, x); by.set_data(0, y)
    best = 1e9
    for _ in range(4):
        for a in al: a.reset()
        sa.timer_start(0); grid.bin(0, al, rows); best = min(best, sa.timer_stop(0))
    res = [np.array(a.get_result()) for a in al]
    if hot: ref2 = res
    ok = np.array_equal(res[0], ref2[0]) and bool(np.all(np.abs(res[2] - ref2[2]) <= 1e-12 * 400.0 * np.maximum(res[0], 1)))
    print(f"hot={hot} 2-D 256x256 count+sum+sum2 (std, 24 B/row): {best:7.3f} ms {rows/best/1e6:6.1f} Grows/s {rows*24/best/1e6/8000:5.3f}  {sa.last_kernel(0)} box {sa.config_get('hot_w')}x{sa.config_get('hot_h')} {sa.config_get('hot_fraction_ppm')/1e4:.1f}% {'same' if ok else 'DIFFERENT'} n={int(res[0].sum())}", flush=True)
sa.config_set("hot", 1)
