#!/bin/bash
# replayed jasper_large step with and without the paired backward launches, alternating, one call
for rep in 1 2 3; do
	for e in 0 auto; do
		l=$(CONVASR_PAIR_BWD=$e CONVASR_NO_PROBE=1 timeout 600 python bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer 2>/dev/null | tail -1)
		echo "pair=$e $rep $(echo $l | python -c 'import json,sys; l=json.load(sys.stdin); print(l["ms_per_step"], l["config"]["whole_step_frac"], (l["config"]["eager_side_stream"] or {}).get("ms_per_step"))')"
	done
done
