#!/bin/bash
# round 4, call B: the new GPU tests, then the configs[4] layer table and the torch-profiler view of one configs[4] step
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4b; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_round4_gpu.py -x -q -s > $O/tests_r4.log 2>&1; echo "r4 tests rc $?"
tail -25 $O/tests_r4.log
timeout 900 python3 scratch/c4_layers.py 376,626,1001 > $O/c4_layers.log 2>&1; tail -4 $O/c4_layers.log
timeout 600 python3 scratch/c4_torchprof.py jasper_large > $O/c4_torchprof.log 2>&1; echo "torchprof rc $?"
timeout 2400 python3 -m pytest tests -x -q -m gpu --deselect tests/test_round4_gpu.py > $O/tests_all.log 2>&1; echo "all tests rc $?"; tail -5 $O/tests_all.log
