#!/bin/bash
# Round-6 measurement artefacts, all from ONE gpurun call (one device).  Raw output under gpurun_out/r06_*, summaries copied to profiles/
# by scratch/summarize_profiles.py r06.
R=${GRAFT_REPO_ROOT:-$PWD}
G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SECONDS=0; python3 $R/bench.py > $G/r06_bench_line.json 2> $G/r06_bench_line.err; echo "default bench.py wall seconds: $SECONDS" > $G/r06_bench_wall.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r06_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg > $G/r06_stats_line.json 2> $G/r06_stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/r06_pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg > /dev/null 2> $G/r06_pmc_$c.log
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $G/r06_pmc_sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg > /dev/null 2> $G/r06_pmc_sq.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $G/r06_pmc_clk -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg > /dev/null 2> $G/r06_pmc_clk.log
# the split-operand path (bf16x3): line, kernel stats, HBM-side traffic of its conv launches
python3 $R/bench.py --dtype bf16x3 --no-cpu-baseline --no-jasper-leg > $G/r06_bench_line_bf16x3.json 2> $G/r06_bench_line_bf16x3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r06_x3_stats -- python3 $R/bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-jasper-leg > $G/r06_x3_stats_line.json 2> $G/r06_x3_stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/r06_x3_pmc_$c -- python3 $R/bench.py --dtype bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --no-jasper-leg > /dev/null 2> $G/r06_x3_pmc_$c.log
done
# configs[4]
python3 $R/bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline > $G/r06_config4_line.json 2> $G/r06_config4_line.err
python3 $R/bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --graph off > $G/r06_config4_line_eager.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r06_config4_stats -- python3 $R/bench.py --workload jasper_large --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer --graph off --side-stream off > $G/r06_config4_stats_line.json 2> $G/r06_config4_stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/r06_config4_pmc_$c -- python3 $R/bench.py --workload jasper_large --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic --graph off --side-stream off > /dev/null 2> $G/r06_config4_pmc_$c.log
done
cd $R
S="--no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 $S > gpurun_out/r06_launcher_n1.json 2> gpurun_out/r06_launcher_n1.err
python3 bench.py $S --steps 20 > gpurun_out/r06_plain_n1.json 2>/dev/null
# one-rank RCCL: the data-parallel step eager and replayed from graphs (collectives captured through rccl.py), both workloads
CONVASR_FORCE_DIST=1 python3 bench.py $S --steps 20 > gpurun_out/r06_rccl_world1.json 2> gpurun_out/r06_rccl_world1.err
CONVASR_FORCE_DIST=1 python3 bench.py $S --steps 20 --graph on > gpurun_out/r06_rccl_world1_graph.json 2> gpurun_out/r06_rccl_world1_graph.err
CONVASR_FORCE_DIST=1 python3 bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --graph off > gpurun_out/r06_rccl_world1_config4.json 2> gpurun_out/r06_rccl_world1_config4.err
CONVASR_FORCE_DIST=1 python3 bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --graph on > gpurun_out/r06_rccl_world1_config4_graph.json 2> gpurun_out/r06_rccl_world1_config4_graph.err
python3 bench.py --dtype f16 --no-cpu-baseline --no-jasper-leg > gpurun_out/r06_bench_line_f16.json 2> gpurun_out/r06_bench_line_f16.err
python3 bench_infer.py > gpurun_out/r06_bench_infer.json 2> gpurun_out/r06_bench_infer.err
rm -f gpurun_out/r06_bench_infer_jasperbig_8k.jsonl
for dt in f32 bf16x3 f16; do for rps in 5 50; do python3 bench_infer.py --model JasperNetBig --sample-rate 8000 --dtype $dt -B 1 -T 6 --rps $rps --duration 10 --no-throughput 2>/dev/null >> gpurun_out/r06_bench_infer_jasperbig_8k.jsonl; done; done
for f in r06_bench_line r06_bench_line_bf16x3 r06_bench_line_f16 r06_launcher_n1 r06_plain_n1 r06_rccl_world1 r06_rccl_world1_graph r06_config4_line r06_config4_line_eager r06_rccl_world1_config4 r06_rccl_world1_config4_graph; do python3 -c "
import json,sys
try:
    j=json.load(open('gpurun_out/$f.json')); print('$f', j['value'], j['ms_per_step'], j['roofline']['frac'] if j.get('roofline') else None, j['config'].get('whole_step_frac'), (j.get('parity') or {}).get('bf16x3_value'), (j.get('dist') or {}).get('exposed_comm_ms'), (j['config'].get('step_graphs') or {}).get('replays'))
except Exception as e: print('$f', 'FAILED', e)
"; done
python3 scratch/summarize_profiles.py r06
