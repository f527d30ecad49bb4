#!/bin/bash
# Round 3, late check on the GPU box: the full -m gpu suite, the differential test's report of where every task part ran, the bench
# line and the kernel stats of the bench command (everything under gpurun_out/r03b/).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b; rm -rf $O; mkdir -p $O
cd $R
VAEX_AMD_REPORT_DIR=$O timeout 420 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_pytest_tail.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu --no-extra --no-configs > $O/bench_prof.json 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 14 | grep -v "at::native\|rocclr" > $O/bench_kernel_stats.txt
rm -rf $O/ks
timeout 120 python $R/tools/r03_skew_groupby.py 2e8 2> /dev/null > $O/skew_groupby.txt
cat $O/gpu_pytest_tail.txt; cat $O/bench_kernel_stats.txt; cat $O/skew_groupby.txt; cut -c1-600 $O/bench.json
