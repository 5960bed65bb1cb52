#!/bin/bash
# Round 6: what do the in-region HIP event pairs around the dominant kernel cost the headline?  Alternating runs on one device.
S="--no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg"
for i in 1 2 3; do
  for v in "events:" "noevents:--no-kernel-timer" "noprobe:--no-kernel-timer" "graph:--graph on --no-kernel-timer"; do
    name=${v%%:*}; flags=${v#*:}
    if [ "$name" = noprobe ]; then export CONVASR_NO_PROBE=1; else unset CONVASR_NO_PROBE; fi
    python3 bench.py $S --steps 20 $flags 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$name', j['ms_per_step'], j['value'])"
  done
done
