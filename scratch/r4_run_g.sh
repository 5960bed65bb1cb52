#!/bin/bash
# round 4, call G: single-pass residual reduce: dense-residual tests + configs[4] line
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4g; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu -k "dense or residual or jasper or bn_ or convblock or config4 or separable or freeze" > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
timeout 900 python3 bench.py --workload jasper_large --steps 8 --warmup 3 --no-cpu-baseline --no-traffic > $O/c4_line.json 2> $O/c4_line.err; python3 -c "
import json; j=json.load(open('$O/c4_line.json')); r=j['roofline']; print('c4', j['value'], j['ms_per_step'], r['frac'], r['wgrad']['frac'], r['whole_step_frac'], {k:(v['launches_per_step'], v['ms_per_step']) for k,v in r['hbm_kernels'].items()})"
