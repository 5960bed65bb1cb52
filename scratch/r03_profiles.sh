#!/bin/bash
# Round-3 measurement artefacts, all from ONE gpurun call (one device).  Raw output under gpurun_out/r03_*, summaries copied to profiles/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r03_bench_line.json 2> $R/gpurun_out/r03_bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $R/gpurun_out/r03_stats_line.json 2> $R/gpurun_out/r03_stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r03_pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic > /dev/null 2> $R/gpurun_out/r03_pmc_$c.log
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/r03_pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic > /dev/null 2> $R/gpurun_out/r03_pmc_sq.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r03_pmc_clk -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic > /dev/null 2> $R/gpurun_out/r03_pmc_clk.log
cd $R
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-traffic > gpurun_out/r03_launcher_n1.json 2> gpurun_out/r03_launcher_n1.err
python3 bench.py --dtype f16 --no-cpu-baseline > gpurun_out/r03_bench_line_f16.json 2> gpurun_out/r03_bench_line_f16.err
python3 bench_infer.py > gpurun_out/r03_bench_infer.json 2> gpurun_out/r03_bench_infer.err
python3 bench.py --no-cpu-baseline --no-traffic > gpurun_out/r03_plain_n1.json 2>/dev/null
for f in r03_bench_line r03_bench_line_f16 r03_launcher_n1 r03_plain_n1; do python3 -c "
import json,sys
try:
    j=json.load(open('gpurun_out/$f.json')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, j.get('dist'))
except Exception as e: print('$f', 'FAILED', e)
"; done
