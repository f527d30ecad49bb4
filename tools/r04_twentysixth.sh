#!/bin/bash
# Round 4, twenty-sixth GPU call (gpurun_out/r04zi/): shared-stream gb_scatter with the exact second attempt (few / skewed keys), dense groupby parts sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zi; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_groupby_fused.py tests/test_gpu_two_ranks.py tests/test_vaex_groupby.py tests/test_gpu_two_procs.py tests/test_vaex_dropin.py tests/test_vaex_differential.py -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head -20
for k in "parts=0" "parts=4" "parts=8"; do timeout 300 python tools/r03_config_one.py c3d 1e9 4 $k > "$O/c3d_$(echo $k | tr ' =' '__').txt" 2>&1; echo "$k"; tail -2 "$O/c3d_$(echo $k | tr ' =' '__').txt" | cut -c1-200; done
timeout 600 python tools/r03_skew_groupby.py 2e8 2>&1 | grep -v amdgpu | cut -c1-250 | head -4
