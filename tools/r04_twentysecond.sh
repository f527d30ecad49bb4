#!/bin/bash
# Round 4, twenty-second GPU call (gpurun_out/r04zd/): heavy keys peeled inside the fused pass (vxh_groupby_run_peeled)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zd; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_groupby_fused.py tests/test_gpu_two_ranks.py tests/test_vaex_groupby.py tests/test_gpu_two_procs.py -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head -20
timeout 600 python tools/r03_skew_groupby.py 2e8 > $O/skew.txt 2>&1; cat $O/skew.txt | cut -c1-240
