#!/bin/bash
# Round 6: SQ counters and effective clock of the split-operand step's kernels (two rocprofv3 --pmc passes over bench.py --dtype bf16x3), one device.
R=${GRAFT_REPO_ROOT:-$PWD}; G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
F="--dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-jasper-leg"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $G/r06_x3_pmc_sq -- python3 $R/bench.py $F > /dev/null 2> $G/r06_x3_pmc_sq.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $G/r06_x3_pmc_clk -- python3 $R/bench.py $F > /dev/null 2> $G/r06_x3_pmc_clk.log
cd $R && python3 scratch/r6_x3_pmc_summary.py
