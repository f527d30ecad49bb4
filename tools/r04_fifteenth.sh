#!/bin/bash
# Round 4, fifteenth GPU call (gpurun_out/r04w/): gb_scatter with branch-free tile loads (K64 / KEEP templates, raw loads settled at the tile's
# turn), gb_reduce software-pipelined across blocks with the compact form as a template, part_scatter_wv without scratch-resident block cursors
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04w; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_groupby_fused.py tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py tests/test_gpu_two_ranks.py tests/test_vaex_groupby.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.txt
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head
for e in 0 1; do
  timeout 300 python tools/r03_config_one.py c3s 1e9 4 gb_early=$e > $O/c3s_early$e.txt 2>&1; tail -3 $O/c3s_early$e.txt
done
timeout 300 python tools/r03_config_one.py c3s 1e9 3 gb_compact=0 > $O/c3s_16byte.txt 2>&1; tail -2 $O/c3s_16byte.txt
timeout 400 python tools/r03_headline_tune.py 1e9 5 wv=3 > $O/ab.txt 2>&1; tail -2 $O/ab.txt
for c in c2 c2e c3d; do timeout 300 python tools/r03_config_one.py $c 1e9 4 > $O/$c.txt 2>&1; tail -2 $O/$c.txt; done
