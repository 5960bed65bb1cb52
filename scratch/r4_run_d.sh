#!/bin/bash
# round 4, call D: fixed r4 tests, JasperNetLarge gradient bars, the limiter probe (600 steps)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4d; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_round4_gpu.py -x -q > $O/tests_r4.log 2>&1; echo "r4 tests rc $?"; tail -5 $O/tests_r4.log
timeout 900 python3 -m pytest tests/test_round2_gpu.py -x -q -s -k "jaspernet_large_dense" > $O/tests_jl.log 2>&1; echo "jl rc $?"; grep "JasperNetLarge 2x5s" $O/tests_jl.log
timeout 900 python3 scratch/limiter_probe.py 600 > $O/limiter.log 2>&1; echo "limiter rc $?"; tail -c 3000 $O/limiter.log
amd-smi metric --help > $O/amd_smi_help.txt 2>&1; amd-smi metric -g 0 > $O/amd_smi_metric.txt 2>&1; amd-smi static -g 0 > $O/amd_smi_static.txt 2>&1
