#!/bin/bash
# the default line and the rocprofv3 kernel statistics of the same command family, one device, final code state
R=${GRAFT_REPO_ROOT:-$PWD}; G=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
SECONDS=0; python3 $R/bench.py > $G/r06_final_line.json 2> $G/r06_final_line.err; echo "default bench.py wall seconds: $SECONDS" > $G/r06_final_wall.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $G/r06_final_stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg > $G/r06_final_stats_line.json 2> $G/r06_final_stats.log
cd $R
cat $G/r06_final_wall.txt
python3 - <<PY
import json, glob, csv
d = json.loads(open('gpurun_out/r06_final_line.json').read().strip().splitlines()[-1]); r = d['roofline']; p = d['parity']
print('line', d['value'], d['ms_per_step'], r['frac'], r['avg_launch_us'], r['launches_per_step'], r.get('events','')[:40], d['config']['whole_step_frac'], r.get('traffic'))
print({k: p[k] for k in p if k.endswith('_ms_per_step')}, d['extra']['jasper_large']['whole_step_frac'])
f = sorted(glob.glob('gpurun_out/r06_final_stats/**/*kernel_stats.csv', recursive = True))[-1]
rows = [x for x in csv.DictReader(open(f)) if 'conv1d_igemm_v2s_kernel' in x['Name']]
for x in rows: print(x['Name'][:90], x['Calls'], round(float(x['AverageNs']) / 1e3, 1))
s = json.loads(open('gpurun_out/r06_final_stats_line.json').read().strip().splitlines()[-1]); print('line under rocprof', s['ms_per_step'], s['roofline']['avg_launch_us'])
PY
