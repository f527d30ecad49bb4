#!/bin/bash
# Round 4, twenty-fourth GPU call (gpurun_out/r04zf/): gb_scatter reading its rows with non-temporal loads ("gb_abl" bit 4), with and without non-temporal
# copy-out stores (bit 3); the reference-test replay after its assertion was brought up to date
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zf; rm -rf $O; mkdir -p $O
cd $R
for k in 0 16 24; do timeout 300 python tools/r03_config_one.py c3s 1e9 4 gb_abl=$k > $O/c3s_abl$k.txt 2>&1; echo "gb_abl=$k"; tail -2 $O/c3s_abl$k.txt | cut -c1-250; done
timeout 900 python -m pytest tests/test_vaex_reference_tests.py -m gpu -q 2>&1 | tail -15 > $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  " $O/pytest.txt | head
