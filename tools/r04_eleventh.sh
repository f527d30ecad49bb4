#!/bin/bash
# Round 4, eleventh GPU call: whole -m gpu suite on the current tree + kernel stats of the bench pass (gpurun_out/r04k/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04k; rm -rf $O; mkdir -p $O
cd $R
VAEX_AMD_REPORT_DIR=$O timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -80 > $O/gpu_pytest_tail.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu --no-extra --no-configs > $O/bench_prof.json 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 14 | grep -v "at::native\|rocclr" > $O/bench_kernel_stats.txt
rm -rf $O/ks
grep -n "passed\|failed" $O/gpu_pytest_tail.txt; grep -n "^E  \|FAILED" $O/gpu_pytest_tail.txt | head -30; cat $O/bench_kernel_stats.txt; cut -c1-900 $O/bench_prof.json
