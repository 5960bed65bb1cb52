#!/bin/bash
# Round-5 measurement artefacts, all from ONE gpurun call (one device).  Raw output under gpurun_out/r05_*, summaries copied to profiles/
# by scratch/summarize_profiles.py r05 (headline) and scratch/summarize_config4.py (configs[4]).
R=${GRAFT_REPO_ROOT:-$PWD}
G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $G/r05_bench_line.json 2> $G/r05_bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r05_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-f16-leg --no-jasper-leg > $G/r05_stats_line.json 2> $G/r05_stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/r05_pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-f16-leg --no-jasper-leg > /dev/null 2> $G/r05_pmc_$c.log
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $G/r05_pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-f16-leg --no-jasper-leg > /dev/null 2> $G/r05_pmc_sq.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $G/r05_pmc_clk -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-f16-leg --no-jasper-leg > /dev/null 2> $G/r05_pmc_clk.log
# configs[4]
python3 $R/bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline > $G/r05_config4_line.json 2> $G/r05_config4_line.err
python3 $R/bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --graph off > $G/r05_config4_line_eager.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r05_config4_stats -- python3 $R/bench.py --workload jasper_large --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer --graph off --side-stream off > $G/r05_config4_stats_line.json 2> $G/r05_config4_stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/r05_config4_pmc_$c -- python3 $R/bench.py --workload jasper_large --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --graph off --side-stream off > /dev/null 2> $G/r05_config4_pmc_$c.log
done
cd $R
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline --no-traffic --no-f16-leg --no-jasper-leg > gpurun_out/r05_launcher_n1.json 2> gpurun_out/r05_launcher_n1.err
CONVASR_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-traffic --no-f16-leg --no-jasper-leg > gpurun_out/r05_rccl_world1.json 2> gpurun_out/r05_rccl_world1.err
python3 bench.py --dtype f16 --no-cpu-baseline --no-jasper-leg > gpurun_out/r05_bench_line_f16.json 2> gpurun_out/r05_bench_line_f16.err
python3 bench_infer.py > gpurun_out/r05_bench_infer.json 2> gpurun_out/r05_bench_infer.err
python3 bench.py --no-cpu-baseline --no-traffic --no-f16-leg --no-jasper-leg > gpurun_out/r05_plain_n1.json 2>/dev/null
python3 bench.py --no-cpu-baseline --no-traffic --no-f16-leg --no-jasper-leg --graph on > gpurun_out/r05_plain_n1_graph.json 2>/dev/null
scratch/r5_run.sh r05_wav2letter trace --workload wav2letter --graph off --side-stream off > gpurun_out/r05_trace_w.log 2>&1
scratch/r5_run.sh r05_jasper_large trace --workload jasper_large --graph off --side-stream off > gpurun_out/r05_trace_j.log 2>&1
python3 scratch/c4_layers.py 376,626,1001 > gpurun_out/r05_c4_layers.log 2>&1
for f in r05_bench_line r05_bench_line_f16 r05_launcher_n1 r05_rccl_world1 r05_plain_n1 r05_plain_n1_graph r05_config4_line r05_config4_line_eager; do python3 -c "
import json,sys
try:
    j=json.load(open('gpurun_out/$f.json')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, j['roofline'].get('whole_step_frac') if j.get('roofline') else None, (j.get('parity') or {}).get('f16_value'), (j.get('dist') or {}).get('exposed_comm_ms'))
except Exception as e: print('$f', 'FAILED', e)
"; done
