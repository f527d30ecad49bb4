#!/bin/bash
# rocprofv3 kernel stats of one command on the GPU box: tools/r03_kstats.sh <tag> <command...>  -> gpurun_out/<tag>_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
O=$R/gpurun_out/ks_$tag; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- "$@" > $R/gpurun_out/${tag}_stdout.txt 2> $O/log.txt
f=$(find $O -name "*kernel_stats.csv" | head -1)
python $R/tools/kstats.py "$f" 16 > $R/gpurun_out/${tag}_kernel_stats.txt
cp $O/log.txt $R/gpurun_out/${tag}_rocprof.log 2>/dev/null; rm -rf $O
tail -5 $R/gpurun_out/${tag}_stdout.txt; cat $R/gpurun_out/${tag}_kernel_stats.txt
