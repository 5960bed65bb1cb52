#!/bin/bash
# round 4, call E: the whole GPU suite on the current tree, W2L torch-profiler view (remaining ATen launches), both bench lines
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4e; mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; echo "all tests rc $?"; tail -6 $O/tests_all.log
timeout 600 python3 scratch/c4_torchprof.py wav2letter > $O/w2l_torchprof.log 2>&1; echo "torchprof rc $?"; grep -E "aten::|Memcpy|Memset" $O/w2l_torchprof.log | cut -c1-60,200-330
timeout 900 python3 bench.py --workload jasper_large --steps 8 --warmup 3 --no-cpu-baseline --no-traffic > $O/c4_line.json 2> $O/c4_line.err; python3 -c "
import json; j=json.load(open('$O/c4_line.json')); r=j['roofline']; print('c4', j['value'], j['ms_per_step'], r['frac'], r['wgrad']['frac'], r['whole_step_frac'])"
timeout 600 python3 bench.py --no-cpu-baseline --no-traffic --no-f16-leg > $O/w2l_line.json 2> $O/w2l.err; python3 -c "
import json; j=json.load(open('$O/w2l_line.json')); r=j['roofline']; print('w2l', j['value'], j['ms_per_step'], r['frac'], r['wgrad']['frac'], r['whole_step_frac'])"
