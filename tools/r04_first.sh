#!/bin/bash
# Round 4, first GPU call: the new packed-counter scenarios, the pool / lazy-fill change on the groupby + finish tests, the bench
# line of this box, and kernel TIMELINES (gaps between launches) of BASELINE configs[2] / [3] — everything under gpurun_out/r04a/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04a; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_groupby_fused.py tests/test_gpu_finish.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest_tail.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
for c in c3d c3s c2; do
  rm -rf $O/kt
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python $R/tools/r03_config_one.py $c 1e9 3 > $O/${c}_stdout.txt 2> $O/${c}_rocprof.log
  f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
  python $R/tools/ktimeline.py "$f" 120 | grep -v "at::native" | tail -60 > $O/${c}_timeline.txt
done
rm -rf $O/kt
cat $O/pytest_tail.txt; for c in c3d c3s c2; do tail -2 $O/${c}_stdout.txt; done; cut -c1-1500 $O/bench.json
