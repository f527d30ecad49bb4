#!/bin/bash
# Round 4, thirty-sixth GPU call (gpurun_out/r04zu/): the wrapped df.groupby uploads plain host columns once (vxh_upload: several copy threads) — tests + the drop-in TIMING lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zu; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_vaex_groupby.py tests/test_vaex_dropin.py tests/test_vaex_differential.py tests/test_vaex_filter.py tests/test_host_surface.py -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -15 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head
grep "groupby" $R/gpurun_out/vaex_dropin_timing.txt | cut -c1-400
python - <<PY
import sys, time, numpy as np
sys.path.insert(0, "$R")
import torch, vaex_amd
sa = vaex_amd.superagg
a = np.random.default_rng(1).integers(0, 1 << 40, 400_000_000, dtype=np.int64)
t = torch.empty(a.shape, dtype=torch.int64, device="cuda")
for th in (1, 2, 4, 6, 8, 12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); sa.upload(a, t, th); dt = time.perf_counter() - t0
    print("upload 3.2 GB with %2d threads: %6.1f ms = %5.1f GB/s" % (th, dt * 1e3, a.nbytes / dt / 1e9), bool((t[::50_000_000].cpu().numpy() == a[::50_000_000]).all()))
t0 = time.perf_counter(); t2 = torch.from_numpy(a).cuda(); torch.cuda.synchronize(); print("torch .cuda(): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
PY
