#!/bin/bash
# Round 4, twenty-first GPU call (gpurun_out/r04zc/): gb_scatter's copy-out with non-temporal stores ("gb_abl" = 8), with and without the early take-over
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zc; rm -rf $O; mkdir -p $O
cd $R
for k in "gb_abl=0" "gb_abl=8" "gb_abl=8 gb_early=1"; do timeout 300 python tools/r03_config_one.py c3s 1e9 4 $k > "$O/c3s_$(echo $k | tr ' =' '__').txt" 2>&1; echo "$k"; tail -2 "$O/c3s_$(echo $k | tr ' =' '__').txt" | cut -c1-250; done
