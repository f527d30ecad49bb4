import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, vaex_amd
sa = vaex_amd.superagg
rows = 1_000_000_000
g = torch.Generator(device="cuda").manual_seed(1234)
x = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g); y = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g)
v = torch.randn(rows, dtype=torch.float64, device="cuda", generator=g) * 2 + 3
torch.cuda.synchronize()
bx = sa.BinnerScalar_float64(1, "x", -4.0, 4.0, 256); by = sa.BinnerScalar_float64(1, "y", -4.0, 4.0, 256)
grid = sa.Grid([bx, by])
aggs = [sa.AggCount_int64(grid, 1, 1), sa.AggSum_float64(grid, 1, 1), sa.AggCount_float64(grid, 1, 1)]
bx.set_data(0, x); by.set_data(0, y); aggs[1].set_data(0, v, 0); aggs[2].set_data(0, v, 0)
for mode in ("reset+bin+result", "reset+bin", "bin only (no reset)"):
    ks, ws = [], []
    for i in range(14):
        t0 = time.perf_counter()
        if mode != "bin only (no reset)":
            for a in aggs: a.reset()
        sa.timer_start(0); grid.bin(0, aggs, rows); ks.append(sa.timer_stop(0))
        if mode == "reset+bin+result":
            r = [a.get_result() for a in aggs]
        ws.append((time.perf_counter() - t0) * 1e3)
    print(mode, "kernel ms:", " ".join(f"{k:.2f}" for k in ks), "| wall ms:", " ".join(f"{w:.2f}" for w in ws), flush=True)
