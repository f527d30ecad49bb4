#!/bin/bash
# Round 4, sixth GPU call: whole -m gpu suite on the new defaults (grouped pass 1, one chunk per 2^30 rows next to a box, reset without a wait) + the bench line (gpurun_out/r04f/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04f; rm -rf $O; mkdir -p $O
cd $R
VAEX_AMD_REPORT_DIR=$O timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/gpu_pytest_tail.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -25 $O/gpu_pytest_tail.txt | cut -c1-250; cut -c1-3000 $O/bench.json
