#!/bin/bash
# Round 4, thirty-first GPU call (gpurun_out/r04zn/): the whole -m gpu suite once more on the last tree (12-byte records for the staged pass 1 behind a knob, off)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zn; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -30 > $O/gpu_pytest_tail.txt
grep -n "passed\|failed" $O/gpu_pytest_tail.txt; grep -n "^E  \|FAILED" $O/gpu_pytest_tail.txt | head -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
