#!/bin/bash
# Round-4 evidence on the GPU box (everything under gpurun_out/r04/, copied into profiles/r04_* by hand): the bench line with its `configs`
# section, rocprofv3 kernel stats of the bench command and of every config, HBM traffic (separate --pmc passes, kernel-trace only: FETCH_SIZE,
# WRITE_SIZE) of the bench pass and of every config, LDS bank-conflict counters of the bench pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 500 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu --no-extra --no-configs > $O/bench_prof.json 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 14 | grep -v "at::native\|rocclr" > $O/bench_kernel_stats.txt
rm -rf $O/ks
for ctr in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$tag -- python $R/bench.py --steps 2 --warmup 1 --no-cpu --no-extra --no-configs --rows 1e9 > /dev/null 2> $O/pmc_$tag.log
done
python $R/tools/pmc_summary.py "$O/pmc_*/*/*counter_collection.csv" > $O/pmc_bench_traffic.txt
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_LDS_BANK_CONFLICT
for c in count2d c2 c2e c3d c3s; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$c -- python $R/tools/r03_config_one.py $c 1e9 3 > $O/${c}_run.txt 2> $O/ks_$c.log
  f=$(find $O/ks_$c -name "*kernel_stats.csv" | head -1)
  (echo "=== $c: $(tail -1 $O/${c}_run.txt)"; python $R/tools/kstats.py "$f" 12 | grep -v "at::native\|rocclr\|fill_kernel") >> $O/configs_kernel_stats.txt
  rm -rf $O/ks_$c
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_${c}_$ctr -- python $R/tools/r03_config_one.py $c 1e9 2 > /dev/null 2> $O/pmc_${c}_$ctr.log
  done
  (echo "=== $c (per dispatch; 2 passes of 1e9 rows)"; python $R/tools/pmc_summary.py "$O/pmc_${c}_*/*/*counter_collection.csv") >> $O/configs_pmc_traffic.txt
  rm -rf $O/pmc_${c}_FETCH_SIZE $O/pmc_${c}_WRITE_SIZE
done
cut -c1-600 $O/bench.json; cat $O/bench_kernel_stats.txt $O/pmc_bench_traffic.txt $O/configs_kernel_stats.txt $O/configs_pmc_traffic.txt
