#!/usr/bin/env python3
"""Round 4: does the bench pass's time depend on WHERE x, y, v sit relative to each other?  Three consecutive processes on one box ran the
same kernel at 5.28 / 5.26 / 4.97 ms (profiles/r04_fresh_vs_second.txt): the columns are torch allocations whose relative offsets differ from
process to process.  Here: one buffer, x / y / v as views at chosen element offsets, every layout timed in ONE process (best of 4, interleaved).
Usage: python tools/r04_alias.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
sa.config_set("wv", int(os.environ.get("WV", "5")))
slack = 1 << 22
buf = torch.empty(3 * rows + 3 * slack, dtype=torch.float64, device="cuda")
g = torch.Generator(device="cuda").manual_seed(1234)
src = [torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(2)]
src.append(torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3)
print("buffer at", hex(buf.data_ptr()), "torch columns at", [hex(t.data_ptr()) for t in src], flush=True)
layouts = [("torch's own three allocations", None)]
for a, b in [(0, 0), (16, 32), (32, 64), (256, 512), (512, 1024), (2048, 4096), (8192, 16384), (65536, 131072), (262144, 524288), (1 << 20, 1 << 21), (528, 1056), (4112, 8224), (65552, 131104)]:
    layouts.append((f"one buffer, y +{a * 8} B, v +{b * 8} B past back-to-back", (a, b)))


def bind(layout):
    if layout is None:
        return src
    a, b = layout
    x = buf[0:rows]; y = buf[rows + a: 2 * rows + a]; v = buf[2 * rows + b: 3 * rows + b]
    for d, s in zip((x, y, v), src):
        d.copy_(s)
    return [x, y, v]

best = {}
ref = None
for rep in range(3):
    for name, layout in layouts:
        x, y, v = bind(layout)
        torch.cuda.synchronize()
        bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
        grid = sa.Grid([bx, by])
        aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
        bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
        for i in range(3):
            for a_ in aggs:
                a_.reset()
            sa.timer_start(0); grid.bin(0, aggs, rows); ms = sa.timer_stop(0)
            if i:
                best[name] = min(best.get(name, 1e9), ms)
        c = np.array(aggs[0].get_result())
        if ref is None:
            ref = c
        assert np.array_equal(c, ref)
for name, _ in layouts:
    print(f"{name:<70} {best[name]:7.3f} ms  {rows * 24 / best[name] / 1e6 / 8000:6.3f} of 8 TB/s", flush=True)
