#!/bin/bash
# Round 4, twenty-third GPU call (gpurun_out/r04ze/): the whole -m gpu suite on the tree with today's kernel rewrites + where every task part of the
# differential calls ran
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04ze; rm -rf $O; mkdir -p $O
cd $R
VAEX_AMD_REPORT_DIR=$O timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/gpu_pytest_tail.txt
grep -n "passed\|failed" $O/gpu_pytest_tail.txt; grep -n "^E  \|FAILED" $O/gpu_pytest_tail.txt | head -30
grep -c "^ok\|^FAIL" $O/differential_report.txt; grep -v "WARNING\|merge used\|here, but\|^ok" $O/differential_report.txt | head -20
