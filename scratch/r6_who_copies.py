"""Which host-side calls of an eager JasperNetLarge training step end in a runtime copy (rocprof: ~100 __amd_rocclr_copyBuffer per step)?  torch.Tensor.copy_ / clone /
contiguous / to are wrapped for one step and the callers counted."""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import convasr_amd as ca
d = torch.device('cuda:0')
torch.manual_seed(1)
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
model = ca.models.JasperNetLarge(64, [38], frontend = fe, dropout = 0.2, check_time_dim_padded = False, compute_dtype = torch.float16).to(d).train()
flat = ca.train.FlatParameters(model)
model._convasr_flat = flat
opt = ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2')
ca.functional.enable_side_stream_wgrad(d, True)
g = torch.Generator().manual_seed(3)
B, T = 16, 16000 * 8
x = (torch.rand(B, T, generator = g) * 2 - 1).to(d); xlen = torch.linspace(0.6, 1, B).to(d)
y = torch.randint(0, 37, (B, 1, 80), generator = g).to(d); ylen = torch.randint(20, 60, (B, 1), generator = g).to(d)
for it in range(3):
	ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = it)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities = [ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack = True) as prof:
	ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = 3)
	torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n = 6)
rows = [e for e in ev if any(t in e.key.lower() for t in ('copy', 'memcpy', 'memset', 'fill', 'zero', 'clone', 'aten::to', '_to_copy'))]
for e in sorted(rows, key = lambda e: -e.count)[:20]:
	st = [l for l in (e.stack or []) if 'convasr_amd' in l][:2]
	print(e.count, e.key, round(e.cuda_time_total if hasattr(e, 'cuda_time_total') else 0, 1), st)
