"""cProfile of the host side of configs[4] training steps (the GPU is drained before each step so that the enqueue never waits)."""
import os, sys, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import convasr_amd as ca
args = bench.parse_args(['--workload', sys.argv[1] if len(sys.argv) > 1 else 'jasper_large', '--steps', '4', '--warmup', '3'])
d = torch.device('cuda:0'); torch.cuda.set_device(d)
torch.manual_seed(1); ca.functional.manual_seed(1)
wl = bench.Workload(args, d, 0, 1)
def step(i):
	x, xlen, y, ylen = wl.batches[i % len(wl.batches)]
	return ca.train.train_step(wl.model, wl.opt, x, xlen, y, ylen, iteration = i)
for i in range(3): step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
for i in range(4):
	torch.cuda.synchronize()
	pr.enable(); step(3 + i); pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream = s).sort_stats('tottime').print_stats(45)
print(s.getvalue())
