#!/bin/bash
# A/B: side-stream operands kept alive until the join instead of record_stream (an allocator event per tensor); JasperNetLarge eager, alternating, one device
for i in 1 2 3; do for v in 0 1; do
  CONVASR_SIDE_KEEPALIVE=$v python bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer --graph off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keepalive=$v', d['ms_per_step'], d['config'].get('whole_step_frac'), d['config'].get('host_enqueue_ms_per_step'), d['config'].get('peak_hbm_gib'))"
done; done
