#!/usr/bin/env python3
"""cProfile of the end-to-end df.groupby (dense int64 keys) to see what surrounds the kernels."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vaex_amd
from vaex_amd.binned import Frame, agg

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 29
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
torch.cuda.synchronize()
ks = (k * 2654435761) % (1 << 40)
df = Frame(dict(v=v, k=k, ks=ks))
KEY = os.environ.get("GROUP_KEY", "k")
spec = {"s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
df.groupby(KEY, spec)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    df.groupby(KEY, spec)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
