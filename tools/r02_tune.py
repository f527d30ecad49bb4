#!/usr/bin/env python3
"""Round-2 A/B of pass 1 of the partition strategy on the BASELINE workloads: part_scatter_wv (barrier-free,
wave-private rings) against part_scatter_blk / part_scatter_f64, with and without the hot box, on N(0,1) and on
uniform x,y; the 3-D 128^3 + selection pass; every variant's result is checked against the first one.
Usage: python tools/r02_tune.py [rows]   (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
xu = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
yu = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
torch.cuda.synchronize()

DEFAULTS = dict(wv=1, wv_waves=8, hot=1, blk=1, strategy=0)


def run_2d(cx, cy, reps=3, **cfg):
    for k, val in dict(DEFAULTS, **cfg).items():
        sa.config_set(k, val)
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256)
    by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    al = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
    al[1].set_data(0, v, 0); al[2].set_data(0, v, 0)
    bx.set_data(0, cx); by.set_data(0, cy)
    best = 1e9
    for _ in range(reps + 1):
        for a in al:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, al, rows)
        best = min(best, sa.timer_stop(0))
    res = [np.array(a.get_result()) for a in al]
    box = f"box {sa.config_get('hot_w')}x{sa.config_get('hot_h')} {sa.config_get('hot_fraction_ppm')/1e4:.1f}%"
    return best, sa.last_kernel(0), box, res


def show(label, ms, kern, extra="", bpr=24):
    print(f"{label:<46} {ms:8.3f} ms {rows/ms/1e6:7.1f} Grows/s {rows*bpr/ms/1e6/8000:6.3f} of 8 TB/s  {kern} {extra}", flush=True)


def same(a, b):
    ok = np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    ok = ok and bool(np.all(np.abs(a[1] - b[1]) <= 1e-12 * 20.0 * np.maximum(a[0], 1)))
    return "same" if ok else "DIFFERENT RESULT"


print(f"rows={rows}")
print("--- N(0,1) x,y: 2-D 256x256 count+sum+count")
ms, k, box, ref = run_2d(x, y, wv=0)
show("old: part_scatter_blk + box", ms, k, box)
for waves in (6, 8, 10):
    ms, k, box, r = run_2d(x, y, wv=2, wv_waves=waves)
    show(f"wv waves={waves} + box", ms, k, box + " " + same(r, ref))
ms, k, box, r = run_2d(x, y, wv=0, hot=0)
show("old, hot=0", ms, k, same(r, ref))
for waves in (8, 12, 16):
    ms, k, box, r = run_2d(x, y, wv=2, wv_waves=waves, hot=0)
    show(f"wv waves={waves}, hot=0", ms, k, same(r, ref))
print("--- uniform x,y")
ms, k, box, refu = run_2d(xu, yu, wv=0)
show("old", ms, k, box)
for waves in (8, 12, 16):
    ms, k, box, r = run_2d(xu, yu, wv=2, wv_waves=waves)
    show(f"wv waves={waves}", ms, k, box + " " + same(r, refu))
for chunk in (1 << 26, 1 << 27):
    ms, k, box, r = run_2d(xu, yu, wv=2, wv_waves=12, part_chunk=chunk)
    show(f"wv waves=12 chunk=2^{chunk.bit_length()-1}", ms, k, same(r, refu))
sa.config_set("part_chunk", 0)

print("--- 3-D 128^3 count with selection (25 B/row)")
del xu, yu
z = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
sel = (v > 3).to(torch.uint8)


def run_3d(reps=2, masked=True, **cfg):
    for k, val in dict(DEFAULTS, **cfg).items():
        sa.config_set(k, val)
    bs = [sa.BinnerScalar_float64(1, nm, -4.0, 4.0, 128) for nm in "xyz"]
    grid = sa.Grid(bs)
    c = sa.AggCount_int64(grid, 1, 1)
    for b, col in zip(bs, (x, y, z)):
        b.set_data(0, col)
    if masked:
        c.set_data_mask(0, sel)
    best = 1e9
    for _ in range(reps + 1):
        c.reset()
        sa.timer_start(0)
        grid.bin(0, [c], rows)
        best = min(best, sa.timer_stop(0))
    return best, sa.last_kernel(0), np.array(c.get_result())


ms, k, ref3 = run_3d(wv=0)
show("old 3-D + selection", ms, k, bpr=25)
for waves in (4, 6, 8):
    ms, k, r = run_3d(wv=2, wv_waves=waves)
    show(f"wv waves={waves} 3-D + selection", ms, k, "same" if np.array_equal(r, ref3) else "DIFFERENT RESULT", bpr=25)
ms, k, ref3n = run_3d(wv=0, masked=False)
show("old 3-D no selection", ms, k)
ms, k, r = run_3d(wv=2, wv_waves=8, masked=False)
show("wv waves=8 3-D no selection", ms, k, "same" if np.array_equal(r, ref3n) else "DIFFERENT RESULT")
for k2, v2 in DEFAULTS.items():
    sa.config_set(k2, v2)
