#!/bin/bash
# Does a process reach the later processes' rate if it simply keeps the GPU busy for a while first?  (gpurun_out/r04p/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04p; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocm-smi --showclocks --showpower > $O/smi0.txt 2>&1
for i in 1 2 3; do
  S=0; [ $i = 1 ] && S=8
  VAEX_AMD_BENCH_SETTLE_S=$S VAEX_AMD_BENCH_STEPS_DEBUG=1 timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-configs --no-cpu --no-extra > $O/bench$i.json 2> $O/bench$i.err
  rocm-smi --showclocks --showpower > $O/smi$i.txt 2>&1
  python - <<PY
import json
d=json.load(open("$O/bench$i.json"))
print("run $i settle $S", round(d['value']/1e9,1), round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['frac'],4), d.get('kernel_ms_per_step'))
print("   trace", d.get('settle_trace'))
PY
done
