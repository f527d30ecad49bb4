import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
import vaex_amd
import bench
sa = vaex_amd.superagg
rows = int(1e9); cpu_rows = int(1e8); shape = 256
gen = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen) * 2 + 3
cb, cpu_res, cpu_rows = bench.cpu_baseline(x, y, v, shape, cpu_rows)
bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, shape)
by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, shape)
grid = sa.Grid([bx, by])
count = sa.AggCount_int64(grid, 1, 1); vsum = sa.AggSum_float64(grid, 1, 1); vcount = sa.AggCount_float64(grid, 1, 1)
aggs = [count, vsum, vcount]
def setd(xx, yy, vv):
    bx.set_data(0, xx); by.set_data(0, yy); bx.clear_data_mask(0); by.clear_data_mask(0)
    vsum.set_data(0, vv, 0); vcount.set_data(0, vv, 0)
    for a in aggs: a.clear_data_mask(0)
def step(n):
    for a in aggs: a.reset()
    grid.bin(0, aggs, n)
    return [a.get_result() for a in aggs]
vmax = float(torch.nan_to_num(v[:cpu_rows]).abs().max().item())
full = None
for it in range(40):
    sa.config_set('hot_cache', it % 2)
    setd(x[:cpu_rows], y[:cpu_rows], v[:cpu_rows])
    g = step(cpu_rows)
    d0 = (g[0] != cpu_res[0]); d2 = (g[2] != cpu_res[2])
    err = np.abs(g[1] - cpu_res[1]); tol = 1e-12 * vmax * np.maximum(g[2], 1)
    bad = d0.any() or d2.any() or (err > tol).any()
    if bad:
        print(it, "SAMPLE BAD", sa.last_kernel(0), "count cells", int(d0.sum()), "delta", (g[0] - cpu_res[0])[d0][:8], "at", np.argwhere(d0)[:8].tolist(), "sum cells", int((err > tol).sum()), (g[1]-cpu_res[1])[err > tol][:8], flush=True)
    setd(x, y, v)
    for k in range(3):
        g = step(rows)
        if full is None: full = [a.copy() for a in g]
        d0 = g[0] != full[0]
        e = np.abs(g[1] - full[1]) > 1e-12 * vmax * np.maximum(g[2], 1)
        if d0.any() or e.any() or (g[2] != full[2]).any():
            print(it, k, "FULL BAD count cells", int(d0.sum()), (g[0] - full[0])[d0][:8], np.argwhere(d0)[:8].tolist(), "sum cells", int(e.sum()), (g[1]-full[1])[e][:8], np.argwhere(e)[:8].tolist(), flush=True)
print("done", int(full[0].sum()))
