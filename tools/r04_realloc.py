#!/usr/bin/env python3
"""Round 4: three processes on one box run the same bench pass at 5.05 / 5.15 / 5.09 ms, each holding its rate to +-0.5 % (and a process
that keeps the GPU busy for 8 s first stays where it started: profiles/r04_fresh_vs_second.txt).  Is the rate a property of WHERE the
columns were allocated?  One process, the three columns allocated anew several times (a pad of varying size allocated first and kept, so
the columns land elsewhere), the pass timed on each.  Usage: python tools/r04_realloc.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
pads_mb = [0, 0, 1, 33, 1000, 4097, 20000, 0]
ref = None
for rep, pad_mb in enumerate(pads_mb):
    torch.cuda.empty_cache()
    pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device="cuda") if pad_mb else None
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    torch.cuda.synchronize()
    bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
    grid = sa.Grid([bx, by])
    aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
    bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
    ms = []
    for i in range(7):
        for a_ in aggs:
            a_.reset()
        sa.timer_start(0); grid.bin(0, aggs, rows); t = sa.timer_stop(0)
        if i >= 2:
            ms.append(t)
    c = np.array(aggs[0].get_result())
    if ref is None:
        ref = c
    assert np.array_equal(c, ref)
    print(f"alloc {rep} pad {pad_mb:6d} MiB  x {hex(x.data_ptr())} y {hex(y.data_ptr())} v {hex(v.data_ptr())}   min {min(ms):.3f} mean {np.mean(ms):.3f} ms  {sa.last_kernel(0)}", flush=True)
    del bx, by, grid, aggs, x, y, v, pad
