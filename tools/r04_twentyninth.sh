#!/bin/bash
# Round 4, twenty-ninth GPU call (gpurun_out/r04zl/): gb_reduce with 8 instead of 4 records per lane per trip (a second library, same box)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04zl; rm -rf $O; mkdir -p $O
cd $R
timeout 300 python tools/r03_config_one.py c3s 1e9 4 > $O/c3s_u4.txt 2>&1; echo U=4; tail -2 $O/c3s_u4.txt | cut -c1-235
cp vaex_amd/lib/libvaexhip.so /tmp/orig.so; cp vaex_amd/lib/libvaexhip_u8.so vaex_amd/lib/libvaexhip.so
timeout 300 python tools/r03_config_one.py c3s 1e9 4 > $O/c3s_u8.txt 2>&1; echo U=8; tail -2 $O/c3s_u8.txt | cut -c1-235
cp /tmp/orig.so vaex_amd/lib/libvaexhip.so
