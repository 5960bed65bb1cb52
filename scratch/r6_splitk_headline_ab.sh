#!/bin/bash
# did the two new kernel parameters (split-K) cost the training headline anything?  HEAD's tree (scratch/_head_tree, built) against the working tree, alternating, one device
F="--steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg"
for i in 1 2 3; do
  for t in scratch/_head_tree .; do
    (cd $t && python bench.py $F 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['roofline'].get('plain_launches',{}).get('avg_launch_us'))"
  done
done
