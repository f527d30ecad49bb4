#!/bin/bash
# Round 4, final evidence on the GPU box: tools/r04_round_profile.sh (bench line with configs, kernel stats, PMC traffic: gpurun_out/r04/), the traffic JSON built
# from its counters, then the whole -m gpu suite with the differential report (gpurun_out/r04/gpu_pytest_tail.txt, differential_report.txt)
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/r04_round_profile.sh > /dev/null 2>&1
O=$R/gpurun_out/r04
python $R/tools/r04_traffic_json.py $O > $O/traffic.json 2> $O/traffic.err
cd $R
VAEX_AMD_REPORT_DIR=$O timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -40 > $O/gpu_pytest_tail.txt
grep -n "passed\|failed" $O/gpu_pytest_tail.txt; grep -n "^E  \|FAILED" $O/gpu_pytest_tail.txt | head -20
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("headline %.4g rows/s  ms/step %.3f  frac %.3f  kernel_ms %.3f  uniform %.4g cold %.4g  %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel_ms"], d.get("value_uniform",0), d.get("value_cold",0), d["config"]["kernel"]))
for c in d.get("configs") or []:
    print("  %-12s %.3g rows/s kernel_ms %.3f frac %.3f ok %s %s" % (c.get("config"), c.get("rows_per_s",0), c.get("kernel_ms",0), c.get("roofline",{}).get("frac",0), (c.get("parity_on_sample") or {}).get("ok"), c.get("groupby_kernels_ms","")))
print("  cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("parity_on_sample"))
t=json.load(open("$O/traffic.json")); print("traffic B/row", t["hbm_bytes_per_row"], {k: (v["hbm_bytes_per_row"], v["ratio"]) for k, v in t["configs"].items()})
PY
cat $O/bench_kernel_stats.txt
