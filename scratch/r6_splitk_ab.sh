#!/bin/bash
# split-K forward for launches of a few tiles: online latency with / without, one device; + the training headline (the kernel gained two parameters)
for m in "Wav2Letter 16000 bf16" "JasperNetBig 8000 f16" "JasperNetBig 8000 bf16x3" "JasperNetBig 8000 bf16"; do set -- $m
  for off in 0 1; do
    CONVASR_NO_SPLITK=$off python bench_infer.py --model $1 --sample-rate $2 --dtype $3 -B 1 -T 6 --rps 20 --duration 6 --no-throughput 2>/dev/null | python -c "
import json,sys
for l in sys.stdin.read().strip().splitlines():
    d=json.loads(l); print('$1 $3 splitk_off=$off graph', d.get('hip_graph'), 'mean', d['mean'], 'median', d['median'], 'service', d.get('service_ms'))"
  done
done
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; done
