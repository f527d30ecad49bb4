#!/bin/bash
# Round-2 evidence on the GPU box (everything under gpurun_out/r02/, copied into profiles/r02_* by hand):
# bench line, rocprofv3 kernel stats of the same command, HBM traffic + SQ counters of the bench pass (separate --pmc passes,
# kernel-trace only), pass-1 A/B and ablations, slab-count scaling, groupby pass knobs, the other BASELINE configs.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu --no-extra > $O/bench_prof.json 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); cp "$f" $O/bench_kernel_stats.csv; python $R/tools/kstats.py $O/bench_kernel_stats.csv 14 | grep -v "at::native\|rocclr" > $O/bench_kernel_stats.txt
rm -rf $O/ks
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$ctr -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --rows 1e9 > /dev/null 2> $O/pmc_$ctr.log
done
python $R/tools/pmc_summary.py "$O/pmc_*/*/*counter_collection.csv" > $O/pmc_bench_traffic.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
PROF_ROWS=1000000000 bash $R/tools/pmc_pass.sh > /dev/null 2>&1; cp $R/gpurun_out/pmc_pass.txt $O/pmc_direct.txt
cd $R
timeout 200 python tools/ab_direct.py 1e9 wv=1 wv=2 wv=3 wv=4 wv=3,wv_waves_direct=8 2>&1 | grep Grows > $O/direct_ab.txt
(echo "# part_scatter_blk (wv=1)"; timeout 100 python tools/ablate_hot.py 1e9 1 2>&1 | grep Grows; echo "# part_scatter_wv DIRECT=1 (wv=3)"; timeout 100 python tools/ablate_hot.py 1e9 3 2>&1 | grep Grows; echo "# part_scatter_wv DIRECT=2 (wv=4)"; timeout 100 python tools/ablate_hot.py 1e9 4 2>&1 | grep Grows) > $O/direct_ablation.txt
timeout 200 python tools/ab_direct.py 1e9 wv=3 wv=3,part_lds=75000 wv=3,part_lds=37000 wv=3,part_lds=18000 wv=4 wv=4,part_lds=75000 2>&1 | grep Grows > $O/direct_streams.txt
timeout 200 python tools/groupby_tune.py 1e9 2>&1 | grep Grows > $O/groupby_tune.txt
timeout 300 python tools/configs_bench.py 1e9 2>/dev/null > $O/configs.txt
timeout 200 python tools/shapes_bench.py 2>&1 | grep Grows > $O/other_shapes.txt
cat $O/bench.json $O/bench_kernel_stats.txt $O/pmc_bench_traffic.txt $O/direct_ab.txt $O/direct_ablation.txt $O/direct_streams.txt $O/groupby_tune.txt $O/configs.txt $O/other_shapes.txt
