#!/bin/bash
# Round 4, thirteenth GPU call: the timed choice between the two pass-1 forms (test), then the bench line (gpurun_out/r04m/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04m; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -30 > $O/pytest.txt
cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py --steps 20 --warmup 5 --no-configs > $O/bench.json 2> $O/bench.err
timeout 400 python $R/bench.py --steps 20 --warmup 5 --no-configs --no-cpu --no-extra > $O/bench2.json 2> $O/bench2.err
grep -n "passed\|failed" $O/pytest.txt; grep -n "^E  \|FAILED" $O/pytest.txt | head; cut -c1-1400 $O/bench.json; cut -c1-600 $O/bench2.json
