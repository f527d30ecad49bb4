import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
import vaex_amd
from oracle import oracle
import bench
sa = vaex_amd.superagg
rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
cpu_rows = int(1e8)
shape = 256
gen = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen)
y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=gen) * 2 + 3
cb, cpu_res, cpu_rows = bench.cpu_baseline(x, y, v, shape, cpu_rows)
print("cpu", cb["value"])
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
def check(tag):
    setd(x[:cpu_rows], y[:cpu_rows], v[:cpu_rows])
    g = step(cpu_rows)
    d0 = int((g[0] != cpu_res[0]).sum()); d2 = int((g[2] != cpu_res[2]).sum())
    err = np.abs(g[1] - cpu_res[1]); tol = 1e-12 * vmax * np.maximum(g[2], 1)
    print(tag, sa.last_kernel(0), "count diff cells", d0, "sum", int(g[0].sum()), int(cpu_res[0].sum()), "countv diff", d2, "sum cells over tol", int((err > tol).sum()), "max err/tol", float((err / tol).max()), flush=True)
check("fresh")
setd(x, y, v); step(rows); step(rows); check("after full steps")
sa.config_set("hot_cache", 0); setd(x, y, v); step(rows); step(rows); sa.config_set("hot_cache", 1); check("after cold steps")
gen_u = torch.Generator(device="cuda").manual_seed(4321)
xu = torch.rand(rows, dtype=torch.float64, device="cuda", generator=gen_u) * 8 - 4
yu = torch.rand(rows, dtype=torch.float64, device="cuda", generator=gen_u) * 8 - 4
setd(xu, yu, v); g = step(rows); print("uniform", sa.last_kernel(0), int(g[0].sum()))
check("after uniform")
for k, val in (("wv", 0), ("hot", 0), ("blk", 0)):
    sa.config_set(k, val); check(f"{k}={val}")
