"""Round 6: where a warmed-up process's first groupby still spends ~80 ms outside the library's kernels — the result columns' page-locked host blocks?
    python tools/r06_first_cols.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import vaex_amd
sa = vaex_amd.superagg
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t0 = time.perf_counter(); sa.warmup(); print("warmup %.1f ms" % ((time.perf_counter() - t0) * 1e3))
n = 200_000_000
g = torch.Generator(device="cuda").manual_seed(1)
k = torch.randint(0, 1_000_000, (n,), dtype=torch.int64, device="cuda", generator=g)
v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); res = sa.groupby_run(k, [v], 2, key_range=(0, 999_999)); t1 = time.perf_counter()
    cols = []
    times = []
    for which, j in ((sa.GB_KEYS, 0), (sa.GB_ROWS, 0), (sa.GB_COUNT, 0), (sa.GB_SUM, 0), (sa.GB_MEAN, 0), (sa.GB_STD, 0)):
        tc = time.perf_counter(); cols.append(np.asarray(res.column(which, j))); times.append(round((time.perf_counter() - tc) * 1e3, 2))
    print("rep %d: groupby_run %.2f ms, columns %s ms, pool mallocs %d (%.1f ms)" % (rep, (t1 - t0) * 1e3, times, sa.config_get("pool_mallocs"), sa.config_get("pool_malloc_us") / 1e3), flush=True)
    del res
    if rep == 1:
        del cols   # (the blocks go back to the host cache)
x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
from vaex_amd.binned import Frame
for rep in range(2):
    t0 = time.perf_counter(); c = Frame(dict(x=x, y=v)).count(binby=["x", "y"], limits=[[-4, 4]] * 2, shape=256); print("count2d rep %d %.2f ms" % (rep, (time.perf_counter() - t0) * 1e3))
