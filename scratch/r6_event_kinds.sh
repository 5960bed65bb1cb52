#!/bin/bash
# What do the in-region event pairs cost with timing-only events (hipEventDisableSystemFence) instead of torch's default ones?  Alternating, one device.
S="--no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg --steps 20"
for i in 1 2 3; do
  for v in "torch:CONVASR_TIMER_EVENTS=torch:" "raw:CONVASR_TIMER_EVENTS=raw:" "raw+dev:CONVASR_TIMER_EVENT_FLAGS=0x60000000:" "dev:CONVASR_TIMER_EVENT_FLAGS=0x40000000:" "none:X=1:--no-kernel-timer"; do
    name=${v%%:*}; rest=${v#*:}; envv=${rest%%:*}; flags=${rest#*:}
    env $envv python3 bench.py $S $flags 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j.get('roofline') or {}
print('$name', j['ms_per_step'], j['value'], r.get('frac'), r.get('avg_launch_us'), (r.get('plain_launches') or {}).get('avg_launch_us'))"
  done
done
