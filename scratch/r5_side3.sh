#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() { python3 bench.py --workload jasper_large --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer "$@" 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['host_enqueue_ms_per_step'], j['loss'])"; }
for rnd in 1 2 3; do
  echo -n "side stream off: "; run --side-stream off
  echo -n "on, branches not launched ahead: "; CONVASR_NO_RES_AHEAD=1 run --side-stream on
  echo -n "on, branches ahead: "; run --side-stream on
done
