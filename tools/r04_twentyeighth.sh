#!/bin/bash
# Round 4, twenty-eighth GPU call (gpurun_out/r04zk/): do the other partition passes gain from streams shared inside an XCD?  The bench pass's grouped cold
# path with small blocks of groups (waves of a region interleave) and 8 regions; 3-D 128^3 and the dense groupby with the new default of >= 8 parts
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zk; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python tools/r03_headline_tune.py 1e9 4 wv_block=1024 wv_block=1024+parts=8 parts=8 wv_block=4096 wv_block=4096+parts=8 wv_block=16384+parts=8 > $O/ab.txt 2>&1; tail -7 $O/ab.txt
for k in "parts=4" "parts=8"; do timeout 300 python tools/r03_config_one.py c2 1e9 4 $k > "$O/c2_$(echo $k | tr ' =' '__').txt" 2>&1; echo "c2 $k"; tail -1 "$O/c2_$(echo $k | tr ' =' '__').txt" | cut -c1-200; done
timeout 300 python tools/r03_config_one.py c3d 1e9 4 > $O/c3d.txt 2>&1; tail -1 $O/c3d.txt
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest.txt; grep -n "passed\|failed" $O/pytest.txt
