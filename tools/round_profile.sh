#!/bin/bash
# Round evidence on the GPU box: bench line, rocprofv3 kernel stats of the same command, HBM traffic counters
# (separate --pmc passes, no trace domains besides kernel-trace), the other shapes/configs.  Output: gpurun_out/prof/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu > $O/bench_prof.json 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats.csv; python $R/tools/kstats.py $O/bench_kernel_stats.csv 14 > $O/bench_kernel_stats.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --rows 5e8 > /dev/null 2> $O/pmc_$ctr.log
done
python $R/tools/pmc_summary.py "$O/pmc_*/*/*counter_collection.csv" > $O/pmc_bench_traffic.txt
timeout 200 python $R/tools/shapes_bench.py 2>&1 | grep Grows > $O/other_shapes.txt
timeout 300 python $R/tools/configs_bench.py 2>&1 | grep Grows > $O/configs.txt
rm -rf $O/ks $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/bench.json $O/bench_kernel_stats.txt $O/pmc_bench_traffic.txt $O/other_shapes.txt $O/configs.txt
