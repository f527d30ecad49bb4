#!/bin/bash
# Round 4, twenty-fifth GPU call (gpurun_out/r04zh/): gb_scatter with record streams shared by the workgroups of an XCD ("gb_sets" = 8), dense groupby with
# fewer / more parts (the slab-partitioned passes share sub-queues between workgroups w % parts already)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zh; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_groupby_fused.py tests/test_gpu_two_ranks.py tests/test_vaex_groupby.py tests/test_gpu_two_procs.py tests/test_vaex_dropin.py -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head -20
for k in "gb_sets=8" "gb_sets=256" "gb_sets=4" "gb_sets=16" "gb_sets=8 gb_compact=0"; do timeout 300 python tools/r03_config_one.py c3s 1e9 4 $k > "$O/c3s_$(echo $k | tr ' =' '__').txt" 2>&1; echo "$k"; tail -1 "$O/c3s_$(echo $k | tr ' =' '__').txt" | cut -c1-235; done
for k in "parts=32" "parts=8" "parts=16" "parts=64"; do timeout 300 python tools/r03_config_one.py c3d 1e9 4 $k > "$O/c3d_$(echo $k | tr ' =' '__').txt" 2>&1; echo "$k"; tail -1 "$O/c3d_$(echo $k | tr ' =' '__').txt" | cut -c1-200; done
timeout 600 python tools/r03_skew_groupby.py 2e8 2>&1 | grep -v amdgpu | cut -c1-200 | head -4
