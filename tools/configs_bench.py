#!/usr/bin/env python3
"""Times the other BASELINE.json configs on one GPU, end to end through vaex_amd.binned.Frame (they are parity-test
cases, not the bench line), and prints one JSON line per config with a `roofline` object (algorithmic bytes per row of
SURVEY §8d / wall time / 8 TB/s):
  configs[2]  3-D histogram (x,y,z) shape 128^3 with a boolean selection mask           25 B/row
  configs[3]  groupby on a 1e6-cardinality int64 key, agg sum/mean/std of v             16 B/row
              dense keys  -> the key column bins itself (ordinal binner), finishers on the device
              scattered   -> the fused radix-partitioned hash aggregation (vxh_groupby_run)
Usage: python tools/configs_bench.py [rows]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import vaex_amd
from vaex_amd.binned import Frame, agg

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(7)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
z = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
sel = v > 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
ks = (k * 2654435761) % (1 << 40)
torch.cuda.synchronize()
df = Frame(dict(x=x, y=y, z=z, v=v, sel=sel, k=k, ks=ks))


def timed(label, config, fn, bytes_per_row, reps=3):
    best = 1e9
    out = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    gbs = rows * bytes_per_row / best / 1e9
    line = {"config": config, "what": label, "rows": rows, "ms": best * 1e3, "rows_per_s": rows / best, "end_to_end": True,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "bytes_per_row": bytes_per_row},
            "kernel": sa.last_kernel(0)}
    info = getattr(df, "last_groupby_info", None)
    if info and "groupby" in label:
        line["groupby_kernels_ms"] = {kk: info[kk] for kk in ("ms_scatter", "ms_reduce", "ms_sort")}
        line["buckets"] = info["buckets"]
    print(json.dumps(line), flush=True)
    return out


c3 = timed("3-D 128^3 count, selection v>3", "configs[2]", lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128, selection="sel"), 25)
assert int(c3.sum()) <= rows
timed("3-D 128^3 count, no selection", "configs[2]'", lambda: df.count(binby=["x", "y", "z"], limits=[[-4, 4]] * 3, shape=128), 24)
spec = {"s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
df.last_groupby_info = None
g1 = timed("groupby dense 1e6 int64 keys: sum/mean/std", "configs[3]", lambda: df.groupby("k", spec), 16, reps=3)
g2 = timed("groupby scattered 1e6 int64 keys: sum/mean/std (fused hash aggregation)", "configs[3]'", lambda: df.groupby("ks", spec), 16, reps=3)
assert len(g1["k"]) == len(g2["ks"])
np.testing.assert_allclose(np.sort(g1["s"]), np.sort(g2["s"]), rtol=1e-9)
timed("2-D 256^2 count+mean incl. host finish", "configs[1]", lambda: df._agg([agg.count(), agg.mean("v")], binby=["x", "y"], limits=[[-4, 4]] * 2, shape=256), 24)
