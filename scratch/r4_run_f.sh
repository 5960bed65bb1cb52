#!/bin/bash
# round 4, call F: remaining GPU tests, then the block-shape A/B (timing + FETCH_SIZE pass) in the diagnostic build
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4f; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests/test_round4_gpu.py -x -q > $O/tests_r4.log 2>&1; echo "r4 tests rc $?"; tail -4 $O/tests_r4.log
export CONVASR_HIP_LIB=$R/convasr_amd/libconvasr_hip.abblock.so
timeout 900 python3 scratch/ab_block.py time > $O/ab_block_time.log 2>&1; echo "ab time rc $?"; cat $O/ab_block_time.log | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ab_block_pmc -- python3 $R/scratch/ab_block.py pmc > $O/ab_block_pmc.log 2>&1; echo "ab pmc rc $?"
cd $R; unset CONVASR_HIP_LIB
python3 scratch/ab_block_summary.py > $O/ab_block_summary.log 2>&1; cat $O/ab_block_summary.log | tail -6
