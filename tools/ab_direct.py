#!/usr/bin/env python3
"""A/B of pass 1 next to the hot box: part_scatter_blk (wv=0/1), part_scatter_wv with rings (wv=2), ring-less (wv=3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
variants = sys.argv[2:] or ["wv=0", "wv=3", "wv=3,wv_waves_direct=12", "wv=3,wv_waves_direct=8", "wv=2"]
g = torch.Generator(device="cuda").manual_seed(1234)
sigma = float(os.environ.get("AB_SIGMA", "1.0"))  # wider columns: a smaller share of the rows inside the hot box
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * sigma
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * sigma
if os.environ.get("AB_UNIFORM"):
    x = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
    y = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
v[::1001] = float("nan")
ref = None
for var in variants:
    cfg = dict(kv.split("=") for kv in var.split(","))
    for k, val in cfg.items(): sa.config_set(k, int(val))
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    al = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
    al[1].set_data(0, v, 0); al[2].set_data(0, v, 0); bx.set_data(0, x); by.set_data(0, y)
    best = 1e9
    for _ in range(4):
        for a in al: a.reset()
        sa.timer_start(0); grid.bin(0, al, rows); best = min(best, sa.timer_stop(0))
    res = [np.array(a.get_result()) for a in al]
    if ref is None: ref = res
    ok = np.array_equal(res[0], ref[0]) and np.array_equal(res[2], ref[2]) and bool(np.all(np.abs(res[1] - ref[1]) <= 1e-12 * 20.0 * np.maximum(res[0], 1)))
    print(f"{var:<34} {best:7.3f} ms {rows/best/1e6:6.1f} Grows/s {rows*24/best/1e6/8000:5.3f}  {sa.last_kernel(0)} box {sa.config_get('hot_w')}x{sa.config_get('hot_h')} {sa.config_get('hot_fraction_ppm')/1e4:.1f}% {'same' if ok else 'DIFFERENT'} n={int(res[0].sum())}", flush=True)
    for k in cfg: sa.config_set(k, {"wv": 3, "wv_waves_direct": 16, "wv_waves": 8, "hot_min_pct": 35, "hot": 1}.get(k, 0))
