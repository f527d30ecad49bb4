#!/bin/bash
# Round 4, eighth GPU call: fused-selection tests again, the scattered groupby under the bucket-load knob, the bench line with configs[2]' (gpurun_out/r04h/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04h; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_selection.py "tests/test_gpu_baseline_shapes.py::test_hot_box_packed_counters_are_exact" tests/test_gpu_groupby_fused.py -m gpu -q 2>&1 | tail -150 > $O/pytest.txt
for l in 50 80 65; do timeout 200 python tools/r03_config_one.py c3s 1e9 4 gb_load_pct=$l 2>&1 | tail -2 >> $O/gb_load.txt; done
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E \|FAILED" $O/pytest.txt | head -30; cat $O/gb_load.txt; cut -c1-1200 $O/bench.json
