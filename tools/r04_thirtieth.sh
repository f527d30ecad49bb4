#!/bin/bash
# Round 4, thirtieth GPU call (gpurun_out/r04zm/): the dense groupby's staged pass 1 with 12-byte AoS records ("f64_rec12" = 1) against the SoA pair
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zm; rm -rf $O; mkdir -p $O
cd $R
for k in "f64_rec12=0" "f64_rec12=1" "f64_rec12=0" "f64_rec12=1"; do timeout 300 python tools/r03_config_one.py c3d 1e9 4 $k >> "$O/c3d_$(echo $k | tr ' =' '__').txt" 2>&1; echo "$k"; tail -2 "$O/c3d_$(echo $k | tr ' =' '__').txt" | cut -c1-200; done
python - <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
import torch, vaex_amd
from vaex_amd.binned import Frame, agg
sa = vaex_amd.superagg
g = torch.Generator(device="cuda").manual_seed(11)
n = 50_000_000
k = torch.randint(0, 1_000_000, (n,), dtype=torch.int64, device="cuda", generator=g)
v = torch.randn(n, dtype=torch.float64, device="cuda", generator=g); v[::777] = float("nan")
spec = {"n": agg.count(), "c": agg.count("v"), "s": agg.sum("v"), "sd": agg.std("v")}
res = {}
for mode in (0, 1):
    sa.config_set("f64_rec12", mode)
    res[mode] = Frame(dict(k=k, v=v)).groupby("k", spec)
    print(mode, sa.last_kernel(0), len(res[mode]["k"]), int(res[mode]["n"].sum()))
for c in ("k", "n", "c"):
    assert np.array_equal(np.asarray(res[0][c]), np.asarray(res[1][c])), c
assert np.allclose(res[0]["s"], res[1]["s"], rtol=1e-12, atol=1e-9) and np.allclose(res[0]["sd"], res[1]["sd"], rtol=1e-9, equal_nan=True)
print("AoS-12 == SoA")
PY
