#!/bin/bash
# Round 4, fifth GPU call: A/B of cheap knobs on the bench pass (waves per workgroup, one chunk, fused merge), the fixed tests (gpurun_out/r04e/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04e; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -k "grouped or packed or config1" tests/test_gpu_two_procs.py -m gpu -q 2>&1 | tail -80 > $O/pytest.txt
timeout 500 python tools/r03_headline_tune.py 1e9 5 merge_fused=0 wv_waves_direct=8 wv_waves_direct=12 part_chunk=536870912 wv_waves_direct=8+part_chunk=536870912 wv=5 wv=5+part_chunk=536870912 wv=5+wv_waves_grouped=6 wv=5+wv_waves_grouped=10 > $O/ab.txt 2>&1
tail -40 $O/pytest.txt | cut -c1-250; cat $O/ab.txt
