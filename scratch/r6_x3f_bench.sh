#!/bin/bash
# the split forward with the one-product backward next to the full split path and the bf16 headline, one device
mkdir -p gpurun_out/x3f
for dt in bf16x3 bf16x3f f16x3f bf16; do
  python bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-jasper-leg --no-parity-legs > gpurun_out/x3f/$dt.json 2> gpurun_out/x3f/$dt.err
  tail -c 600 gpurun_out/x3f/$dt.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/x3f/$dt.json').read().strip().splitlines()[-1])
r = d.get('roofline') or {}
print('$dt', d['ms_per_step'], d['value'], 'parity', (d.get('parity') or {}).get('ctc_loss_rel_err'), 'frac', r.get('frac'), 'wgrad', (r.get('wgrad') or {}).get('ms_per_step'), 'dgrad1', (r.get('dgrad_one_product') or {}).get('ms_per_step'), 'main ms', r.get('ms_per_step'), 'stack', (r.get('conv_stack') or {}))
print({k: v.get('ms_per_step') for k, v in (r.get('hbm_kernels') or {}).items()})
PY
done
