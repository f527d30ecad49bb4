#!/bin/bash
# Round 4, second GPU call: microbench6's new modes, the new tests, then the whole -m gpu suite (gpurun_out/r04b/)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04b; rm -rf $O; mkdir -p $O
cd $R
timeout 120 tools/microbench6 1e9 new > $O/microbench6.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_two_procs.py "tests/test_gpu_parity.py::test_vxh_allreduce_native_world1" "tests/test_gpu_parity.py::test_rccl_allreduce_path_single_rank" "tests/test_gpu_baseline_shapes.py::test_hot_box_packed_counters_are_exact" -m gpu -q 2>&1 | tail -25 > $O/pytest_new.txt
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_all.txt
cat $O/microbench6.txt; cat $O/pytest_new.txt; cat $O/pytest_all.txt
