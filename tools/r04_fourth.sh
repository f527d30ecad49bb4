#!/bin/bash
# Round 4, fourth GPU call: grouped pass 1 — parity tests again, ablations of its cold path, pass-2 variants (gpurun_out/r04d/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04d; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -k "grouped or packed" -m gpu -q 2>&1 | tail -60 > $O/pytest_grouped.txt
timeout 600 python -m pytest tests/test_gpu_two_procs.py tests/test_vaex_arrow_columns.py tests/test_vaex_dropin.py tests/test_vaex_differential.py tests/test_vaex_groupby.py -m gpu -q 2>&1 | tail -150 > $O/pytest_new.txt
timeout 400 python tools/r03_headline_tune.py 1e9 4 wv=5 wv=5+no_pipeline=2 wv=5+no_pipeline=256 wv=5+part_chunk=536870912 part_chunk=536870912 wv=5+parts=16 wv=5+parts=64 > $O/ab_grouped.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rm -rf $O/ks
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/tools/r03_headline_tune.py 1e9 3 wv=5 > $O/ks_wv5.out 2> $O/ks_wv5.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 10 | grep -v "at::native\|rocclr" > $O/kernel_stats_wv5.txt
rm -rf $O/ks
tail -30 $O/pytest_grouped.txt | cut -c1-250; tail -60 $O/pytest_new.txt | cut -c1-300; cat $O/ab_grouped.txt; cat $O/kernel_stats_wv5.txt
