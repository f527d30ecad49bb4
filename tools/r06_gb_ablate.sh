#!/bin/bash
# Round 6: where gb_scatter's time goes (the ABLATION library: results wrong on purpose).  gb_abl bits: 1 copy-out computes but does not store, 2 no copy-out,
# 4 no staging and no copy-out, 8 non-temporal copy-out stores, 16 no position atomics (nothing staged or copied), 32 no row loads (synthetic keys)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export LD_LIBRARY_PATH=$R/tools/ablate:$LD_LIBRARY_PATH
for cfg in c3s c3; do
  for abl in 0 1 2 4 20 32 36 52 8; do
    echo "== $cfg gb_abl=$abl"
    python tools/r03_config_one.py $cfg 1e9 3 gb_abl=$abl 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids" | tail -1 | sed -e 's/gb_scatter+gb_reduce//' | cut -c1-330
  done
done
