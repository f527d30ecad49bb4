#!/usr/bin/env python3
"""Round 3 (late): df.groupby on SKEWED scattered int64 keys (Zipf 1.3 clipped at 4e5 distinct values: the head of the law and the
clip value hold ~60 % of the rows), device-resident: with the heavy-hitter peel (Frame._groupby_peeled) and without it (the fused
pass alone overflows its spare blocks and the call falls back to ordered_set + BinnerHash).  Usage: r03_skew_groupby.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vaex_amd
from vaex_amd.binned import Frame, agg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
rng = np.random.default_rng(3)
z = rng.zipf(1.3, rows // 8)
k = torch.from_numpy((np.minimum(z, 400_000) * 2654435761) % (1 << 40)).cuda().repeat(8)
k = k[torch.randperm(len(k), device="cuda")] if rows <= 400_000_000 else k
v = torch.randn(len(k), dtype=torch.float64, device="cuda") * 2 + 3
torch.cuda.synchronize()
spec = {"c": agg.count("v"), "s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
uni = (torch.randint(0, 1_000_000, (len(k),), dtype=torch.int64, device="cuda") * 2654435761) % (1 << 40)
kd = torch.from_numpy(np.minimum(z, 1_000_000)).cuda().repeat(8)       # the same law on a DENSE key range (1e6 cells: the slab-partitioned pass)
kd = kd[torch.randperm(len(kd), device="cuda")] if rows <= 400_000_000 else kd
for label, keys, peel in (("zipf keys, peeled", k, True), ("zipf, 3-pass peel (r3)", k, 3), ("zipf keys, no peel", k, False), ("uniform 1e6 keys", uni, True),
                          ("dense zipf, peeled", kd, True), ("dense zipf, no peel", kd, False), ("dense uniform 1e6", uni // 2654435761 % 1_000_000 if False else torch.randint(0, 1_000_000, (len(k),), dtype=torch.int64, device="cuda"), True)):
    df = Frame(dict(k=keys, v=v))
    if not peel:
        df.heavy_key_rows = 1 << 62
    if peel == 3:   # (round 3's peel: ordinals + keep-mask + a dense groupby of the heavy rows; round 4 peels inside the fused pass)
        df.one_kernel_peel = False
    best = 1e9
    for rep in range(3):
        df.last_groupby_info = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = df.groupby("k", spec)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    info = df.last_groupby_info or {}
    print(f"{label:<22} {len(keys):.3g} rows  {best*1e3:9.2f} ms  {len(keys)/best/1e9:7.2f} Grows/s  groups {len(res['k'])}  rows counted {int(res['c'].sum())}  info {({k_: info[k_] for k_ in ('buckets','retries','heavy_keys','heavy_groups','dense','ms_scatter','ms_reduce','ms_sort') if k_ in info})}", flush=True)
