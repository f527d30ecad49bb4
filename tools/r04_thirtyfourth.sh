#!/bin/bash
# Round 4, thirty-fourth GPU call (gpurun_out/r04zr/): tiles dealt region-wise per XCD ("wv_xcd" = 1) on the bench pass and on 3-D 128^3
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zr; rm -rf $O; mkdir -p $O
cd $R
timeout 400 python tools/r03_headline_tune.py 1e9 5 wv_xcd=1 wv_xcd=1+wv_span=4 wv_xcd=1+wv_span=64 wv=3 wv=3+wv_xcd=1 > $O/ab.txt 2>&1; tail -6 $O/ab.txt
for k in "wv_xcd=0" "wv_xcd=1"; do timeout 300 python tools/r03_config_one.py c2 1e9 4 $k > "$O/c2_$(echo $k | tr ' =' '__').txt" 2>&1; echo "c2 $k"; tail -2 "$O/c2_$(echo $k | tr ' =' '__').txt" | cut -c1-200; done
