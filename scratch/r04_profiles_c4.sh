#!/bin/bash
# The configs[4] half of scratch/r04_profiles.sh alone (after a change that touches only that workload).
R=${GRAFT_REPO_ROOT:-$PWD}
G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline > $G/r04_config4_line.json 2> $G/r04_config4_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r04_config4_stats -- python3 $R/bench.py --workload jasper_large --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer > $G/r04_config4_stats_line.json 2> $G/r04_config4_stats.log
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/r04_config4_pmc_$c -- python3 $R/bench.py --workload jasper_large --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic > /dev/null 2> $G/r04_config4_pmc_$c.log
done
cd $R
python3 bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --side-stream off > gpurun_out/r04_config4_line_no_side_stream.json 2>/dev/null
for f in r04_config4_line r04_config4_line_no_side_stream; do python3 -c "
import json
j=json.load(open('gpurun_out/$f.json')); print('$f', j['value'], j['ms_per_step'], j['config']['side_stream_wgrad'], j['config']['host_enqueue_ms_per_step'], j['roofline']['frac'], j['roofline']['whole_step_frac'])"; done
