import sys, os, time; sys.path.insert(0, '.')
import numpy as np, torch
import vaex_amd
from vaex_amd.binned import Frame, agg
sa = vaex_amd.superagg
rows = int(1e9)
g = torch.Generator(device="cuda").manual_seed(7)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
k = torch.randint(0, 1_000_000, (rows,), dtype=torch.int64, device="cuda", generator=g)
df = Frame(dict(v=v, k=k))
spec = {"s": agg.sum("v"), "m": agg.mean("v"), "sd": agg.std("v")}
descs = list(spec.values())
for _ in range(2): df.groupby("k", spec)
def sync(): sa.synchronize() if hasattr(sa, "synchronize") else None; torch.cuda.synchronize()
for rep in range(3):
    sync(); t0 = time.perf_counter()
    kmin, kmax = df._key_range("k", k, "int64"); count = kmax - kmin + 1
    binby = [dict(column="k", count=count, min_value=kmin)]
    t1 = time.perf_counter()
    specs, grid, aggs, want = df._pass(descs + [agg.count()], binby)
    t2 = time.perf_counter(); sync(); t3 = time.perf_counter()
    fin = [d.finish_spec(sa, [aggs[i] for i in ids]) for d, ids in zip(descs, want[:-1])]
    cols, index = sa.finish(fin, present=aggs[want[-1][0]], first=0, n=count, want_index=True)
    t4 = time.perf_counter()
    out = [np.asarray(c) for c in cols] + [np.asarray(index) + kmin]
    t5 = time.perf_counter()
    print(f"range {1e3*(t1-t0):.2f}  pass(enqueue) {1e3*(t2-t1):.2f}  pass(wait) {1e3*(t3-t2):.2f}  finish {1e3*(t4-t3):.2f}  to numpy {1e3*(t5-t4):.2f}  total {1e3*(t5-t0):.2f} ms", flush=True)
