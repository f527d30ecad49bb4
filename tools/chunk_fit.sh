#!/bin/bash
# kernel durations of the partition pair for several chunk sizes (rocprofv3 kernel trace) -> fixed vs per-row cost
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for ch in ${CHUNKS:-25 26 27 28 29}; do
  rm -rf $R/gpurun_out/kt; mkdir -p $R/gpurun_out/kt
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -- python $R/tools/prof_pass.py 536870912 part_chunk=$((1<<ch)) > $R/gpurun_out/kt/log.txt 2>&1
  f=$(find $R/gpurun_out/kt -name "*kernel_trace.csv" | head -1)
  python - "$f" $ch <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "part_" in n:
        d["scatter" if "scatter" in n else "reduce"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print(f"chunk=2^{sys.argv[2]} {k:8s} n={len(v):3d} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f} us")
PY
done
rm -rf $R/gpurun_out/kt
