#!/bin/bash
# Round 4, ninth GPU call: compact-record groupby (tests + timing), two-rank heavy-key peel, groupby suites (gpurun_out/r04i/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04i; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_groupby_fused.py tests/test_gpu_two_ranks.py tests/test_vaex_groupby.py tests/test_gpu_finish.py -m gpu -q 2>&1 | tail -150 > $O/pytest.txt
for kv in gb_compact=1 gb_compact=0; do timeout 200 python tools/r03_config_one.py c3s 1e9 5 $kv 2>&1 | tail -3 >> $O/gb_compact.txt; done
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E \|FAILED" $O/pytest.txt | head -40; cat $O/gb_compact.txt
