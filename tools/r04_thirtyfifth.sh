#!/bin/bash
# Round 4, thirty-fifth GPU call (gpurun_out/r04zt/): the grouped flush ranks its 64 records by slab with two DPP wave scans of byte counters instead of eight ballots
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zt; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_selection.py -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -12 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head
timeout 300 python tools/r03_headline_tune.py 1e9 5 wv=3 > $O/ab.txt 2>&1; tail -2 $O/ab.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python $R/bench.py --no-cpu --no-extra --no-configs > $O/bench_prof.json 2> $O/ks.log
f=$(find $O/ks -name "*kernel_stats.csv" | head -1); python $R/tools/kstats.py "$f" 8 | grep -v "at::native\|rocclr\|fill_kernel"
rm -rf $O/ks
