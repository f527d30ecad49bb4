#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04o; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/r04_alias.py 1e9 > $O/alias.txt 2>&1
cat $O/alias.txt
