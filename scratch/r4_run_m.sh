#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4m; mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; echo "all tests rc $?"; tail -4 $O/tests_all.log
timeout 900 python3 bench.py --workload jasper_large --steps 8 --warmup 3 --no-cpu-baseline --no-traffic > $O/c4_line.json 2> $O/c4_line.err; python3 -c "
import json; j=json.load(open('$O/c4_line.json')); r=j['roofline']; print('c4', j['value'], j['ms_per_step'], r['frac'], r['wgrad']['frac'], r['whole_step_frac'], {k:(v['launches_per_step'], v['ms_per_step']) for k,v in r['hbm_kernels'].items()})"
timeout 600 python3 bench.py --no-cpu-baseline --no-traffic --no-f16-leg > $O/w2l_line.json 2> $O/w2l.err; python3 -c "
import json; j=json.load(open('$O/w2l_line.json')); r=j['roofline']; print('w2l', j['value'], j['ms_per_step'], r['frac'], r['launches_per_step'], r['wgrad']['frac'], r['whole_step_frac'], list(r['hbm_kernels']))"
