#!/bin/bash
# usage: ab_many.sh ROUNDS NAME...   -- bench.py alternately with each library variant (NAME 'nofuse' = main library with CONVASR_NO_BWD_FUSION=1)
rounds=$1; shift
for r in $(seq $rounds); do for n in "$@"; do
  lib=convasr_amd/libconvasr_hip.$n.so; nf=0
  if [ "$n" = main ]; then lib=convasr_amd/libconvasr_hip.so; fi
  if [ "$n" = nofuse ]; then lib=convasr_amd/libconvasr_hip.so; nf=1; fi
  CONVASR_NO_BWD_FUSION=$nf CONVASR_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-traffic --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; pl=r.get('plain_launches')
n=r['launches_per_step']; tot=r['avg_launch_us']*n
fused=(tot-pl['avg_launch_us']*pl['launches_per_step'])/(n-pl['launches_per_step']) if pl else 0
print('%-12s ms %.3f  v2s: %d launches avg %.1f us (plain %.1f, fused-epilogue %.1f)  wgrad_ms %.3f  stack_ms %.3f  reduce_ms %.3f apply_ms %.3f' % ('$n', j['ms_per_step'], n, r['avg_launch_us'], pl['avg_launch_us'] if pl else r['avg_launch_us'], fused, r['wgrad']['ms_per_step'], r['conv_stack']['ms_per_step'], r['hbm_kernels']['bn_act_bwd_reduce_kernel']['ms_per_step'], r['hbm_kernels']['bn_act_bwd_apply_kernel']['ms_per_step']))"
done; done
