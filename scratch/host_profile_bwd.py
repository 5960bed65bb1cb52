"""cProfile of the BACKWARD half of the host side (it runs on autograd's device thread, where the main thread's profiler does not see it):
a profiler is enabled from inside the first backward function that runs on that thread."""
import os, sys, cProfile, pstats, io, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import convasr_amd as ca
from convasr_amd import functional as Fn
args = bench.parse_args(['--workload', sys.argv[1] if len(sys.argv) > 1 else 'jasper_large', '--steps', '4', '--warmup', '3'])
d = torch.device('cuda:0'); torch.cuda.set_device(d)
torch.manual_seed(1); ca.functional.manual_seed(1)
wl = bench.Workload(args, d, 0, 1)
prof = {}
active = [False]
orig = Fn.CtcLossFunction.backward  # the first backward function of a step
def first_backward(ctx, g):
	if active[0]:
		t = threading.get_ident()
		if t not in prof:
			prof[t] = cProfile.Profile()
		prof[t].enable()
	return orig(ctx, g)
Fn.CtcLossFunction.backward = staticmethod(first_backward)
def step(i):
	x, xlen, y, ylen = wl.batches[i % len(wl.batches)]
	r = ca.train.train_step(wl.model, wl.opt, x, xlen, y, ylen, iteration = i)
	for p in prof.values(): p.disable()
	return r
for i in range(3): step(i)
torch.cuda.synchronize()
active[0] = True
for i in range(4):
	torch.cuda.synchronize(); step(3 + i)
torch.cuda.synchronize()
for t, p in prof.items():
	s = io.StringIO()
	pstats.Stats(p, stream = s).sort_stats('tottime').print_stats(40)
	print('thread', t, s.getvalue())
