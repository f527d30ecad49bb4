#!/bin/bash
# which kernels run for the dense groupby with f64_rec12 = 1 (kernel names under rocprofv3)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zo; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/tools/r03_config_one.py c3d 1e9 3 f64_rec12=1 > $O/run.txt 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 8 | grep -v "at::native\|rocclr\|fill_kernel"
rm -rf $O/ks
