"""Upper bound of what folding the BN finalize launches into their producers could give: the bench step with the 36 finalize launches
per step skipped after warm-up (stale statistics: timing only)."""
import sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import _lib
skip = os.environ.get('X_SKIP') == '1'
from convasr_amd import ops
orig = ops.call
count = {'n': 0}
def call(name, *a):
	if skip and name in ('convasr_bn_finalize', 'convasr_bn_bwd_finalize'):
		count['n'] += 1
		if count['n'] > 36 * 3: return 0
	return orig(name, *a)
ops.call = call
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-traffic', '--no-kernel-timer']
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'), run_name = '__main__')
