#!/bin/bash
# Round 4, thirty-seventh GPU call (gpurun_out/r04zv/): last whole -m gpu suite + smoke on the final tree (concurrent column uploads in the wrapped df.groupby)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zv; rm -rf $O; mkdir -p $O
cd $R
VAEX_AMD_REPORT_DIR=$O timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/gpu_pytest_tail.txt
grep -n "passed\|failed" $O/gpu_pytest_tail.txt; grep -n "^E  \|FAILED" $O/gpu_pytest_tail.txt | head -20
grep "groupby" $R/gpurun_out/vaex_dropin_timing.txt | cut -c1-330
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
