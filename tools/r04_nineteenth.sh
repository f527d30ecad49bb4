#!/bin/bash
# Round 4, nineteenth GPU call (gpurun_out/r04za/): mask bytes kept raw in part_scatter_wv (c2), and the scatter's per-row time against the row count
# (the per-workgroup write window shrinks with the rows: a TLB / page-locality bound would show as a faster small run)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04za; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py tests/test_gpu_selection.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head
for c in c2 c2e; do timeout 300 python tools/r03_config_one.py $c 1e9 4 > $O/$c.txt 2>&1; tail -2 $O/$c.txt; done
for n in 1.25e8 2.5e8 5e8 1e9; do timeout 300 python tools/r03_config_one.py c3s $n 3 > $O/c3s_$n.txt 2>&1; tail -1 $O/c3s_$n.txt; done
