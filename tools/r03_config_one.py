#!/usr/bin/env python3
"""One BASELINE config, a few passes on device-resident columns — for rocprofv3 (kernel stats / --pmc FETCH_SIZE, WRITE_SIZE).
Usage: python tools/r03_config_one.py c2|c2e|count2d|uni|bench|c3d|c3s [rows] [passes] [knob=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vaex_amd
from vaex_amd.binned import Frame, agg
sa = vaex_amd.superagg
which = sys.argv[1]
rows = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000_000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for kv in sys.argv[4:]:   # knobs: key=value
    sa.config_set(kv.split("=")[0], int(kv.split("=")[1]))
g = torch.Generator(device="cuda").manual_seed(7)
if which == "c2":
    x, y, z, v = (torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(4))
    sel = (v * 2 + 3 > 3).to(torch.uint8)
    del v
    df = Frame(dict(x=x, y=y, z=z, sel=sel))
    run = lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128, selection="sel", edges=True)
elif which == "c2e":   # (round 4) the selection as an expression over a fourth float64 column: evaluated inside the binning kernel
    x, y, z, v = (torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(4))
    v = v * 2 + 3
    df = Frame(dict(x=x, y=y, z=z, v=v))
    run = lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128, selection="v > 3", edges=True)
elif which in ("uni", "bench"):   # (round 6) the bench pass — count(*), sum(v), count(v) on 256 x 256 — over UNIFORM x, y (no dense box) / over N(0,1)
    mk = (lambda: torch.rand(rows, dtype=torch.float64, device="cuda", generator=g) * 8 - 4) if which == "uni" else (lambda: torch.randn(rows, dtype=torch.float64, device="cuda", generator=g))
    x, y = mk(), mk()
    v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    df = Frame(dict(x=x, y=y, v=v))
    run = lambda: df._agg([agg.count(), agg.mean("v")], binby=["x", "y"], limits=[[-4, 4]] * 2, shape=256, edges=True)
elif which == "count2d":   # (round 4) north_star's target sentence: 2-D count(*) on a 256 x 256 grid
    x, y = (torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(2))
    df = Frame(dict(x=x, y=y))
    run = lambda: df.count(binby=["x", "y"], limits=[[-4, 4]] * 2, shape=256, edges=True)
else:
    v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
    if which in ("c3s", "c3s2"):
        k = (k * 2654435761) % (1 << 40)
    if which in ("c3w", "c3w2"):   # (round 6) 1e6 distinct keys spread over the whole int64 range (hashed ids): no compact records
        k = k * 0x9E3779B97F4A7C15 + 12345
    df = Frame(dict(k=k, v=v))
    spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
    if which in ("c3s2", "c3w2", "c3d2"):   # (round 6) two value columns
        df = Frame(dict(k=k, v=v, w=torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)))
        spec = {"c": agg.count("v"), "m": agg.mean("v"), "sw": agg.sum("w"), "sdw": agg.std("w")}
    run = lambda: df.groupby("k", spec)
torch.cuda.synchronize()
for i in range(passes):
    t0 = time.perf_counter(); sa.timer_start(0); r = run(); k_ms = sa.timer_stop(0); dt = time.perf_counter() - t0
    print(f"{which} pass {i}: {dt*1e3:.3f} ms wall, {k_ms:.3f} ms on the stream = {rows/dt/1e9:.1f} Grows/s  {sa.last_kernel(0)} {getattr(df, 'last_groupby_info', '') if which.startswith('c3') else ''}", flush=True)
