#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rnd in 1 2; do
for ss in "" "--side-stream"; do
  python3 bench.py --workload jasper_large --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer $ss 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4', '$ss', j['value'], j['ms_per_step'], j['config']['host_enqueue_ms_per_step'])"
done; done
for ss in "" "--side-stream"; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg $ss 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w2l', '$ss', j['value'], j['ms_per_step'])"
done
