#!/bin/bash
# round 4, call J: the whole GPU suite + smoke + the default bench line on the current tree
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4j; mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests_all.log 2>&1; echo "all tests rc $?"; tail -4 $O/tests_all.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"; python3 -c "
import json; j=json.load(open('$O/bench_line.json')); r=j['roofline']; print('w2l', j['value'], j['ms_per_step'], r['frac'], r['traffic'], r['wgrad']['frac'], r['whole_step_frac'], j['parity']['f16_value'], j['parity']['ctc_loss_rel_err'], j['cpu_baseline']['value'])"
