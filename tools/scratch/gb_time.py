import sys, os, time; sys.path.insert(0, '.')
import numpy as np, torch, cProfile, pstats
import vaex_amd
from vaex_amd.binned import Frame, agg
rows = int(1e9)
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
df = Frame(dict(v=v, k=k))
spec = {"s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
for _ in range(2): df.groupby("k", spec)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); out = df.groupby("k", spec); torch.cuda.synchronize(); t1 = time.perf_counter()
pr.disable()
print("groupby ms", (t1 - t0) * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
