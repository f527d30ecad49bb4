#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04j; rm -rf $O; mkdir -p $O
cd $R
for kv in gb_compact=1 gb_compact=0; do timeout 200 python tools/r03_config_one.py c3s 1e9 4 $kv 2>&1 | tail -3 >> $O/gb_compact.txt; done
cat $O/gb_compact.txt
