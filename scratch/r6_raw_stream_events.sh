#!/bin/bash
# A/B: the side-stream hand-overs of the eager JasperNetLarge step (one per weight gradient) through torch's events vs fence-free raw events; alternating, one device
for i in 1 2 3; do for v in 0 1; do
  CONVASR_RAW_STREAM_EVENTS=$v python bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer --graph off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('raw_stream_events=$v', d['ms_per_step'], d['config'].get('whole_step_frac'))"
done; done
