#!/bin/bash
# replayed vs eager after the memset / memcpy nodes left the captured step: configs[4] and the Wav2Letter headline, one device
for mode in on off on off; do
  python bench.py --workload jasper_large --steps 12 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer --graph $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('jasper_large graph $mode', d['ms_per_step'], d['config'].get('whole_step_frac'), d['config'].get('host_ms_per_step'))"
done
for mode in on off on off; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg --no-parity-legs --no-jasper-leg --graph $mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wav2letter graph $mode', d['ms_per_step'], d['config'].get('whole_step_frac'))"
done
