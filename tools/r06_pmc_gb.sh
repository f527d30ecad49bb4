#!/bin/bash
# SQ / LDS counters of gb_scatter / gb_reduce for one config of tools/r03_config_one.py (three --pmc passes, kernel-trace only).
#   bash tools/r06_pmc_gb.sh <out name> <config> [knob=value ...]      -> gpurun_out/<out name>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$1; CFG=$2; shift 2
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmcgb; mkdir -p /tmp/pmcgb $R/gpurun_out
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmcgb/s$i -- python $R/tools/r03_config_one.py $CFG 1e9 2 "$@" > /tmp/pmcgb/log$i.txt 2>&1
done
cd $R; (echo "# $CFG $@"; python tools/pmc_summary.py "/tmp/pmcgb/s*/*/*counter_collection.csv") > gpurun_out/$OUT.txt; cat gpurun_out/$OUT.txt
