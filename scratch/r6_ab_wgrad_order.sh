#!/bin/bash
# Round 6: the weight gradient on the side stream ORDERED AFTER its layer's dgrad (the next layer's BN passes run under it) against the serial step and
# against round 3's concurrent form; alternating runs on one device.
S="--no-cpu-baseline --no-traffic --no-f16-leg --no-parity-legs --no-jasper-leg --no-kernel-timer --steps 20"
run() { python3 bench.py $S "$@" 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'])"; }
for i in 1 2 3; do
  echo "serial $(run --side-stream off)"
  echo "concurrent(r3) $(run --side-stream on)"
  echo "after_dgrad $(CONVASR_WGRAD_AFTER_DGRAD=1 run --side-stream on)"
  echo "after_dgrad+no_bwd_fusion $(CONVASR_WGRAD_AFTER_DGRAD=1 CONVASR_NO_BWD_FUSION=1 run --side-stream on)"
done
