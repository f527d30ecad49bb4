#!/bin/bash
# Round 2: per-kernel durations + SQ counters + HBM traffic of the bench pass under ONE configuration.
# Usage (GPU box): bash tools/r02_prof_one.sh tag [key=value ...]   e.g.  bash tools/r02_prof_one.sh direct wv=3
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
O=$R/gpurun_out/r02_$tag; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/tools/prof_pass.py 1e9 "$@" > $O/run.txt 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1)
echo "=== $@: $(tail -1 $O/run.txt)" > $O/kernel_stats.txt
python $R/tools/kstats.py "$f" 8 | grep -v "at::native\|rocclr\|fill_kernel" >> $O/kernel_stats.txt
rm -rf $O/ks
if [ -z "$NO_PMC" ]; then
PROF_ROWS=1000000000 bash $R/tools/pmc_pass.sh "$@" > /dev/null 2>&1; cp $R/gpurun_out/pmc_pass.txt $O/pmc.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/t_$ctr -- python $R/tools/prof_pass.py 1e9 "$@" > /dev/null 2> $O/t_$ctr.log
done
python $R/tools/pmc_summary.py "$O/t_*/*/*counter_collection.csv" > $O/traffic.txt
rm -rf $O/t_FETCH_SIZE $O/t_WRITE_SIZE
fi
cat $O/kernel_stats.txt; cat $O/pmc.txt $O/traffic.txt 2>/dev/null | grep -A19 "part_scatter"
