#!/bin/bash
# Does the device-state probe (hwmon reads every 5 ms from a native thread) cost the step anything?  And the side stream on Wav2Letter?  One call, alternating.
tag=$1
common="--steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg --no-jasper-leg --graph off"
for rep in 1 2 3; do
	for v in probe noprobe side r04; do
		case $v in
		probe) l=$(timeout 600 python bench.py $common 2>/dev/null | tail -1) ;;
		noprobe) l=$(CONVASR_NO_PROBE=1 timeout 600 python bench.py $common 2>/dev/null | tail -1) ;;
		side) l=$(CONVASR_NO_PROBE=1 timeout 600 python bench.py $common --side-stream on 2>/dev/null | tail -1) ;;
		r04) l=$(cd scratch/_r04_tree && timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg 2>/dev/null | tail -1) ;;
		esac
		echo "$v $rep $(echo $l | python -c 'import json,sys; l=json.load(sys.stdin); print(l["ms_per_step"], l["value"])')"
	done
done | tee gpurun_out/${tag}_ab_probe.txt
