#!/bin/bash
# Round 4, twelfth GPU call: grouped tests after the header-line change, pass-2 ablation, kernel stats (gpurun_out/r04l/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04l; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_fuzz.py -k "grouped or packed or fuzz or config1" -m gpu -q 2>&1 | tail -30 > $O/pytest.txt
timeout 400 python tools/r03_headline_tune.py 1e9 5 no_pipeline=8192 wv=3 > $O/ab.txt 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/tools/r03_headline_tune.py 1e9 3 no_pipeline=8192 > $O/ks.out 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 10 | grep -v "at::native\|rocclr" > $O/kernel_stats.txt
rm -rf $O/ks
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head; tail -4 $O/ab.txt; cat $O/kernel_stats.txt
