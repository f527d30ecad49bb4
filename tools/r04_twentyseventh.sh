#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zj; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_groupby_fused.py tests/test_gpu_two_ranks.py tests/test_vaex_groupby.py -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head -20
