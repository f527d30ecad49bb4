#!/bin/bash
# Round 4, seventeenth GPU call (gpurun_out/r04y/): the bench pass on UNIFORM x, y under the pass-1 choices (box + staged queues, box-less ring kernel,
# box + ring-less / grouped cold path forced at a 27 % box)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04y; rm -rf $O; mkdir -p $O
cd $R
VAEX_TUNE_DIST=uniform timeout 500 python tools/r03_headline_tune.py 1e9 4 hot=0 hot_direct_pct=10 hot_direct_pct=10+wv=3 hot=0+wv_span=1 hot=0+wv_waves=16 > $O/uniform.txt 2>&1
cat $O/uniform.txt
