#!/usr/bin/env python3
"""One BASELINE config under several knob settings in ONE process, interleaved, best of N (boxes and processes differ by several per cent;
only lines of one run compare).  Usage: python tools/config_ab.py c2|c2e|count2d|c3d|c3s rows reps setting [setting ...]
  setting: "-" (defaults) or knob=value[+knob=value...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vaex_amd
from vaex_amd.binned import Frame, agg
sa = vaex_amd.superagg
which, rows, reps = sys.argv[1], int(float(sys.argv[2])), int(sys.argv[3])
settings = [({} if a == "-" else {p.split("=")[0]: int(p.split("=")[1]) for p in a.split("+")}) for a in sys.argv[4:]] or [{}]
g = torch.Generator(device="cuda").manual_seed(7)
bytes_per_row = {"c2": 25, "c2e": 32, "count2d": 16, "c3d": 16, "c3s": 16}[which]
if which in ("c2", "c2e"):
    x, y, z, v = (torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(4))
    v = v * 2 + 3
    if which == "c2":
        df = Frame(dict(x=x, y=y, z=z, sel=(v > 3).to(torch.uint8)))
        del v
        run = lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128, selection="sel", edges=True)
    else:
        df = Frame(dict(x=x, y=y, z=z, v=v))
        run = lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128, selection="v > 3", edges=True)
elif which == "count2d":
    x, y = (torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) for _ in range(2))
    df = Frame(dict(x=x, y=y))
    run = lambda: df.count(binby=["x", "y"], limits=[[-4, 4]] * 2, shape=256, edges=True)
else:
    v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
    k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
    if which == "c3s":
        k = (k * 2654435761) % (1 << 40)
    df = Frame(dict(k=k, v=v))
    spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
    run = lambda: df.groupby("k", spec)
torch.cuda.synchronize()
best = [1e9] * len(settings)
ref = None
for r in range(reps + 1):
    for i, cfg in enumerate(settings):
        saved = {k: sa.config_get(k) for k in cfg}
        for k, val in cfg.items():
            sa.config_set(k, val)
        sa.timer_start(0); res = run(); sa.timer_stop(0); k_ms = sa.timer_kernels_ms(0)
        for k, val in saved.items():
            sa.config_set(k, val)
        if r:
            best[i] = min(best[i], k_ms)
        else:
            print(cfg, sa.last_kernel(0), flush=True)
        chk = np.asarray(res if not isinstance(res, dict) else res["c"])
        if ref is None:
            ref = chk
        else:
            assert np.array_equal(chk, ref), cfg
for cfg, ms in zip(settings, best):
    print(f"{which} {str(cfg):<44} {ms:8.3f} ms kernels  {rows/ms/1e6:7.1f} Grows/s  {rows*bytes_per_row/ms/1e6/8000:6.3f} of 8 TB/s on {bytes_per_row} B/row", flush=True)
