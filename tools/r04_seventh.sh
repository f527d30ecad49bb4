#!/bin/bash
# Round 4, seventh GPU call: fused selections (class-level + through real vaex), the two tests that failed in the full run (gpurun_out/r04g/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04g; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python -m pytest "tests/test_gpu_baseline_shapes.py::test_hot_box_packed_counters_are_exact" tests/test_gpu_two_procs.py -m gpu -q 2>&1 | tail -120 > $O/pytest_failed.txt
timeout 900 python -m pytest tests/test_gpu_selection.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_vaex_dropin.py tests/test_vaex_differential.py tests/test_vaex_filter.py tests/test_vaex_named_selection.py tests/test_golden_api.py -m gpu -q 2>&1 | tail -120 > $O/pytest_sel.txt
grep -n "passed\|failed" $O/pytest_failed.txt $O/pytest_sel.txt; grep -n "^E " $O/pytest_failed.txt | head -30; grep -n "^E \|FAILED" $O/pytest_sel.txt | head -40
