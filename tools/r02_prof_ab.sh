#!/bin/bash
# Round 2: per-kernel durations of the bench pass under four pass-1 variants, SQ counters and HBM traffic of one of them.
# Usage (GPU box): bash tools/r02_prof_ab.sh [pmc key=value ...]   e.g.  bash tools/r02_prof_ab.sh wv=1 hot=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_ab; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cfg in "wv=1" "wv=0" "wv=1 hot=0" "wv=0 hot=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$tag -- python $R/tools/prof_pass.py 1e9 $cfg > $O/run_$tag.txt 2> $O/ks_$tag.log
  f=$(find $O/ks_$tag -name "*kernel_stats.csv" | head -1)
  echo "=== $cfg: $(tail -1 $O/run_$tag.txt)" >> $O/kernel_stats.txt
  python $R/tools/kstats.py "$f" 8 | grep -v "at::native\|rocclr\|fill_kernel" >> $O/kernel_stats.txt
  rm -rf $O/ks_$tag
done
PROF_ROWS=1000000000 bash $R/tools/pmc_pass.sh "$@" > /dev/null 2>&1; cp $R/gpurun_out/pmc_pass.txt $O/pmc_$(echo "$@" | tr ' =' '__').txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/t_$ctr -- python $R/tools/prof_pass.py 1e9 "$@" > /dev/null 2> $O/t_$ctr.log
done
python $R/tools/pmc_summary.py "$O/t_*/*/*counter_collection.csv" > $O/traffic_$(echo "$@" | tr ' =' '__').txt
rm -rf $O/t_FETCH_SIZE $O/t_WRITE_SIZE
cat $O/kernel_stats.txt; cat $O/pmc_*.txt; cat $O/traffic_*.txt
