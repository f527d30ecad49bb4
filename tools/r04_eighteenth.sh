#!/bin/bash
# Round 4, eighteenth GPU call (gpurun_out/r04z/): part_reduce_fast with the record form as a template and trips pipelined across blocks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04z; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_shapes.py tests/test_gpu_selection.py tests/test_golden_api.py tests/test_reference_kat.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head
for c in c2 c3d; do timeout 300 python tools/r03_config_one.py $c 1e9 4 > $O/$c.txt 2>&1; tail -2 $O/$c.txt; done
VAEX_TUNE_DIST=uniform timeout 300 python tools/r03_headline_tune.py 1e9 4 hot=0 > $O/uniform.txt 2>&1; tail -2 $O/uniform.txt
timeout 300 python tools/r03_headline_tune.py 1e9 4 wv=3 > $O/ab.txt 2>&1; tail -2 $O/ab.txt
