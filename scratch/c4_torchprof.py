"""Which host-side ops of a configs[4] step launch ATen kernels / memcpys (torch.profiler, one step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
args = bench.parse_args(['--workload', sys.argv[1] if len(sys.argv) > 1 else 'jasper_large', '--steps', '1', '--warmup', '2'])
import convasr_amd as ca
d = torch.device('cuda:0'); torch.cuda.set_device(d)
wl = bench.Workload(args, d, 0, 1)
def step(i):
	x, xlen, y, ylen = wl.batches[i % len(wl.batches)]
	return ca.train.train_step(wl.model, wl.opt, x, xlen, y, ylen, iteration = i)
for i in range(3): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities = [ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack = True) as prof:
	step(3); torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n = 6).table(sort_by = 'self_cuda_time_total', row_limit = 60, max_name_column_width = 60, max_src_column_width = 110))
