#!/bin/bash
# Round 4, third GPU call: the grouped pass 1 — parity tests, A/B against the ring-less pass 1 in one process, per-kernel times (gpurun_out/r04c/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -k "grouped or packed" tests/test_gpu_two_procs.py tests/test_vaex_arrow_columns.py -m gpu -q 2>&1 | tail -150 > $O/pytest_new.txt
timeout 300 python tools/r03_headline_tune.py 1e9 4 wv=5 wv=5+wv_waves_grouped=12 wv=5+wv_waves_grouped=16 wv=5+no_pipeline=64 no_pipeline=64 > $O/ab_grouped.txt 2>&1
cd /tmp; export TMPDIR=/tmp
for w in 3 5; do
  rm -rf $O/ks
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/tools/r03_headline_tune.py 1e9 3 wv=$w > $O/ks_wv$w.out 2> $O/ks_wv$w.log
  f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 10 | grep -v "at::native\|rocclr" > $O/kernel_stats_wv$w.txt
done
rm -rf $O/ks
tail -40 $O/pytest_new.txt | cut -c1-250; cat $O/ab_grouped.txt; cat $O/kernel_stats_wv3.txt $O/kernel_stats_wv5.txt
