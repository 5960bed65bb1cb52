#!/bin/bash
# Round 6: A/B of one-optimisation-off hooks on the headline step, alternating runs on one device.  Usage: scratch/r6_ab_env.sh VAR1=1 VAR2=1 ...
S="--no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg --no-kernel-timer --steps 20"
for i in 1 2 3; do
  python3 bench.py $S 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('default', j['ms_per_step'])"
  for v in "$@"; do
    env $v python3 bench.py $S 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'])"
  done
done
