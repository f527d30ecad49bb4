#!/bin/bash
# Is the first process on a fresh box slower than the second?  The same bench command three times in a row, per-step kernel times printed (gpurun_out/r04n/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04n; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for i in 1 2 3; do
  VAEX_AMD_BENCH_STEPS_DEBUG=1 timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-configs --no-cpu --no-extra > $O/bench$i.json 2> $O/bench$i.err
  python - <<PY
import json
d=json.load(open("$O/bench$i.json"))
print("run $i", round(d['value']/1e9,1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), d.get('kernel_ms_per_step'))
PY
done
rocm-smi --showclocks 2>/dev/null | head -20
