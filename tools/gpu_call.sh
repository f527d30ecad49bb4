#!/bin/bash
# ONE parametrised runner for a gpurun call (round 5; replaces the per-call r04_<ordinal>.sh scripts).
#   gpurun --timeout T -- 'bash tools/gpu_call.sh <name> <step> [<step> ...]'
# Every step's output goes to gpurun_out/<name>/<step>.txt (merged back by gpurun); a step is one of
#   tests[:<pytest args>]      python -m pytest -m gpu -q <args, default: tests>          (':' separates, ',' -> ' ')
#   smoke                      __graft_entry__.smoke()
#   bench[:<bench.py args>]    python bench.py <args>  -> bench.json
#   kstats[:<bench.py args>]   rocprofv3 --kernel-trace --stats of bench.py --no-cpu --no-extra --no-configs <args> -> kernel_stats.txt
#                              and the bench line of THAT SAME process -> bench_profiled.json
#   pmc[:<set>;<set>...]       one rocprofv3 --pmc pass per counter set (',' inside a set; kernel-trace only, never with other traces) of a short bench command -> pmc_<n>.txt
#   cfgprof[:<configs>]        kernel stats + FETCH_SIZE / WRITE_SIZE passes of every BASELINE config alone -> configs_kernel_stats.txt, configs_pmc_traffic.txt
#   tune:<args>                python tools/r03_headline_tune.py <args>
#   py:<script>[:<args>]       python <script> <args>
#   sh:<command>               bash -c <command>
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
O=$R/gpurun_out/$NAME; rm -rf $O; mkdir -p $O
cd $R
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
filter() { grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl\|^$"; }
n=0
for step in "$@"; do
  n=$((n+1))
  kind=${step%%:*}; rest=""; [[ "$step" == *:* ]] && rest=${step#*:}
  args=${rest//,/ }
  echo "== [$n] $step" | tee -a $O/steps.txt
  t0=$(date +%s)
  case $kind in
    tests)  VAEX_AMD_REPORT_DIR=$O timeout 1500 python -m pytest -m gpu -q ${args:-tests} 2>&1 | filter | tail -40 > $O/tests_$n.txt
            grep -n "passed\|failed\|error" $O/tests_$n.txt | tail -3; grep -n "^E  \|FAILED" $O/tests_$n.txt | head -20 ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt ;;
    bench)  timeout 900 python bench.py $args 2> $O/bench_$n.err | tail -1 > $O/bench_$n.json
            python tools/bench_summary.py $O/bench_$n.json ;;
    kstats) rm -rf /tmp/prof_$n
            (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -- python $R/bench.py --no-cpu --no-extra --no-configs $args 2> $O/kstats_$n.err | tail -1 > $O/bench_profiled_$n.json)
            f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); python tools/kstats.py "$f" 14 | grep -v "at::native\|rocclr" > $O/kernel_stats_$n.txt 2>&1; cat $O/kernel_stats_$n.txt
            python tools/bench_summary.py $O/bench_profiled_$n.json $O/kernel_stats_$n.txt ;;
    pmc)    i=0; IFS=';' read -ra sets <<< "${rest:-FETCH_SIZE;WRITE_SIZE}"
            for ctr in "${sets[@]}"; do i=$((i+1)); rm -rf /tmp/pmc_${n}_$i
              (cd /tmp && timeout 600 rocprofv3 --pmc ${ctr//,/ } --kernel-trace --output-format csv -d /tmp/pmc_${n}_$i -- python $R/bench.py --no-cpu --no-extra --no-configs --steps 2 --warmup 1 > /dev/null 2> $O/pmc_${n}_$i.err)
            done
            python tools/pmc_summary.py "/tmp/pmc_${n}_*/*/*counter_collection.csv" > $O/pmc_$n.txt 2>&1; head -60 $O/pmc_$n.txt ;;
    cfgprof) # per-config kernel stats + HBM traffic counters (tools/r03_config_one.py <config> 1e9): configs_kernel_stats.txt, configs_pmc_traffic.txt
            for c in ${args:-count2d c2 c2e c3d c3s}; do
              rm -rf /tmp/ks_$c; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$c -- python $R/tools/r03_config_one.py $c 1e9 3 > $O/${c}_run.txt 2> $O/ks_$c.err)
              f=$(find /tmp/ks_$c -name "*kernel_stats.csv" | head -1)
              (echo "=== $c: $(tail -1 $O/${c}_run.txt)"; python tools/kstats.py "$f" 12 | grep -v "at::native\|rocclr\|fill_kernel") >> $O/configs_kernel_stats.txt
              for ctr in FETCH_SIZE WRITE_SIZE; do
                rm -rf /tmp/pmc_${c}_$ctr; (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${c}_$ctr -- python $R/tools/r03_config_one.py $c 1e9 2 > /dev/null 2> $O/pmc_${c}_$ctr.err)
              done
              (echo "=== $c (per dispatch; 2 passes of 1e9 rows)"; python tools/pmc_summary.py "/tmp/pmc_${c}_*/*/*counter_collection.csv") >> $O/configs_pmc_traffic.txt
            done
            cat $O/configs_kernel_stats.txt | head -60 ;;
    tune)   timeout 1200 python tools/r03_headline_tune.py $args 2>&1 | filter | tee $O/tune_$n.txt | tail -30 ;;
    py)     script=${rest%%:*}; a=""; [[ "$rest" == *:* ]] && a=${rest#*:}; timeout 1500 python $script ${a//,/ } 2>&1 | filter | tee $O/py_$n.txt | tail -40 ;;
    sh)     timeout 1500 bash -c "$rest" 2>&1 | filter | tee $O/sh_$n.txt | tail -40 ;;
    *)      echo "unknown step $step" ;;
  esac
  echo "   ($(( $(date +%s) - t0 )) s)" | tee -a $O/steps.txt
done
