#!/usr/bin/env python3
"""Round 3: the bench pass (2-D 256x256 count+sum+count, 1e9 N(0,1) rows) under a few knobs in ONE process, interleaved
(boxes drift): best of N per setting.  Usage: python tools/r03_headline_tune.py [rows] [reps] [key=v1,v2,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
settings = [dict()]
for kv in sys.argv[3:]:
    if "+" in kv:   # one setting of several knobs: a=1+b=2
        settings.append({p.split("=")[0]: int(p.split("=")[1]) for p in kv.split("+")})
        continue
    k, vs = kv.split("=")
    settings += [{k: int(v)} for v in vs.split(",")]
g = torch.Generator(device="cuda").manual_seed(1234)
if os.environ.get("VAEX_TUNE_DIST") == "uniform":   # (round 4) bench.py's value_uniform data: x, y ~ U(-4, 4)
    x = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
    y = torch.rand(rows, dtype=torch.float64, device="cuda", generator=g) * 8 - 4
else:
    x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
    y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
torch.cuda.synchronize()
bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
grid = sa.Grid([bx, by])
aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
best = [1e9] * len(settings)
ref = None
for r in range(reps + 1):
    for i, cfg in enumerate(settings):
        saved = {k: sa.config_get(k) for k in cfg}
        for k, val in cfg.items():
            sa.config_set(k, val)
        for a in aggs:
            a.reset()
        sa.timer_start(0)
        grid.bin(0, aggs, rows)
        ms = sa.timer_stop(0)
        if r:
            best[i] = min(best[i], ms)
        else:
            print(cfg, "box", sa.config_get("hot_w"), "x", sa.config_get("hot_h"), "fraction", sa.config_get("hot_fraction_ppm") / 1e4, "%", sa.last_kernel(0), flush=True)
        res = [np.array(a.get_result()) for a in aggs]
        if ref is None:
            ref = res
        elif "no_pipeline" not in cfg:   # (ablation bits drop work on purpose)
            assert np.array_equal(res[0], ref[0]) and np.array_equal(res[2], ref[2]) and np.all(np.abs(res[1] - ref[1]) <= 1e-12 * 20 * np.maximum(ref[0], 1)), cfg
        for k, val in saved.items():
            sa.config_set(k, val)
for cfg, ms in zip(settings, best):
    print(f"{str(cfg):<40} {ms:8.3f} ms {rows/ms/1e6:7.1f} Grows/s {rows*24/ms/1e6/8000:6.3f} of 8 TB/s", flush=True)
