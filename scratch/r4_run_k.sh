#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4k; mkdir -p $O
cd $R
export CONVASR_HIP_LIB=$R/convasr_amd/libconvasr_hip.wgorder.so
timeout 900 python3 scratch/ab_wgrad_order.py time > $O/time.log 2>&1; echo "time rc $?"; tail -5 $O/time.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc -- python3 $R/scratch/ab_wgrad_order.py pmc > $O/pmc.log 2>&1; echo "pmc rc $?"
