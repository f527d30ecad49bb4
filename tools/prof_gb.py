#!/usr/bin/env python3
"""The fused hash groupby (vxh_groupby_run) on scattered 1e6-cardinality int64 keys, a few runs — for rocprofv3.
Usage: python tools/prof_gb.py [rows]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import vaex_amd

sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = (torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g) * 2654435761) % (1 << 40)
torch.cuda.synchronize()
for _ in range(3):
    res = sa.groupby_run(k, [v], 2)
print(rows, len(res), res.info())
