#!/bin/bash
# round 4, call C: tail128 + K=1 flatten + K=1 in-place weights: tests, configs[4] line, layer table
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4c; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_round4_gpu.py -x -q > $O/tests_r4.log 2>&1; echo "r4 tests rc $?"; tail -5 $O/tests_r4.log
timeout 2400 python3 -m pytest tests -x -q -m gpu --deselect tests/test_round4_gpu.py > $O/tests_all.log 2>&1; echo "all tests rc $?"; tail -5 $O/tests_all.log
timeout 900 python3 bench.py --workload jasper_large --steps 8 --warmup 3 --no-cpu-baseline --no-traffic > $O/c4_line.json 2> $O/c4_line.err; python3 -c "
import json; j=json.load(open('$O/c4_line.json')); r=j['roofline']; print('c4', j['value'], j['ms_per_step'], r['frac'], r['wgrad']['frac'], r['whole_step_frac'], {k:(v['ms_per_step']) for k,v in r['hbm_kernels'].items()})"
timeout 900 python3 scratch/c4_layers.py 376,626 > $O/c4_layers.log 2>&1; grep totals $O/c4_layers.log
timeout 600 python3 bench.py --no-cpu-baseline --no-traffic --no-f16-leg > $O/w2l_line.json 2> $O/w2l.err; python3 -c "
import json; j=json.load(open('$O/w2l_line.json')); r=j['roofline']; print('w2l', j['value'], j['ms_per_step'], r['frac'], r['wgrad']['frac'], r['whole_step_frac'])"
