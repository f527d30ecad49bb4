#!/bin/bash
# SQ counters of the kernels of the BASELINE pass (tools/prof_pass.py), three --pmc passes; output gpurun_out/pmc_pass.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1)); timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc/s$i -- python $R/tools/${PROF_SCRIPT:-prof_pass.py} ${PROF_ROWS:-536870912} "$@" > $R/gpurun_out/pmc/log$i.txt 2>&1
done
cd $R; python tools/pmc_summary.py "gpurun_out/pmc/s*/*/*counter_collection.csv" > gpurun_out/pmc_pass.txt; rm -rf gpurun_out/pmc; cat gpurun_out/pmc_pass.txt
