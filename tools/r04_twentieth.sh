#!/bin/bash
# Round 4, twentieth GPU call (gpurun_out/r04zb/): where gb_scatter's 8.25 ms go — its phases switched off one by one ("gb_abl": results wrong on purpose)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zb; rm -rf $O; mkdir -p $O
cd $R
for a in 0 1 2 4; do timeout 300 python tools/r03_config_one.py c3s 1e9 3 gb_abl=$a > $O/c3s_abl$a.txt 2>&1; echo "gb_abl=$a"; tail -1 $O/c3s_abl$a.txt | cut -c1-260; done
