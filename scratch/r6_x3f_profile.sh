#!/bin/bash
# 'bf16x3f' (split forward, one-product backward): the default bench line (with the new parity leg), the stand-alone line, rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-$PWD}
G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SECONDS=0; python3 $R/bench.py > $G/r06_bench_line_v2.json 2> $G/r06_bench_line_v2.err; echo "default bench.py wall seconds: $SECONDS" > $G/r06_bench_wall_v2.txt
python3 $R/bench.py --dtype bf16x3f --no-cpu-baseline --no-jasper-leg > $G/r06_bench_line_bf16x3f.json 2> $G/r06_bench_line_bf16x3f.err
python3 $R/bench.py --dtype bf16x3f --graph on --no-cpu-baseline --no-jasper-leg --no-traffic > $G/r06_bench_line_bf16x3f_graph.json 2> $G/r06_bench_line_bf16x3f_graph.err
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r06_x3f_stats -- python3 $R/bench.py --dtype bf16x3f --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-jasper-leg > $G/r06_x3f_stats_line.json 2> $G/r06_x3f_stats.log
cd $R
cat $G/r06_bench_wall_v2.txt
for f in r06_bench_line_v2 r06_bench_line_bf16x3f r06_bench_line_bf16x3f_graph; do python3 - <<PY
import json
try:
    d = json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
    p = d.get('parity') or {}
    print('$f', d['dtype'], d['ms_per_step'], d['value'], 'frac', (d.get('roofline') or {}).get('frac'), {k: p[k] for k in p if k.endswith('_ms_per_step') or k.endswith('_value') or k.endswith('_error')}, p.get('ctc_loss_rel_err'))
except Exception as e:
    print('$f', 'ERR', e); print(open('gpurun_out/$f.err').read()[-1500:])
PY
done
find $G/r06_x3f_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {} | cut -c1-160
