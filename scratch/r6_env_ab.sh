#!/bin/bash
# runtime environment switches, alternating, one device: the bf16 headline (20 steps), the JasperNetLarge eager step, one online request
S="--no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg --no-kernel-timer --steps 20"
for i in 1 2; do for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0"; do
  env $e python3 bench.py $S 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline $e', j['ms_per_step'])"
  env $e python3 bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer --graph off 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('jasper_large eager $e', j['ms_per_step'], j['config'].get('host_enqueue_ms_per_step'))"
  env $e python3 bench_infer.py --model JasperNetBig --sample-rate 8000 --dtype f16 -B 1 -T 6 --rps 20 --duration 5 --no-throughput 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin.read().strip().splitlines():
    d=json.loads(l); print('infer $e graph', d['hip_graph'], d['mean'], d['service_ms'])"
done; done
