"""Round 6 debug: the adamw_bf16 interleave test without per-step synchronisation; prints AdamW's applied-step rows per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import convasr_amd as ca
d = torch.device('cuda:0')
MODE = sys.argv[1] if len(sys.argv) > 1 else 'plain'

def batch(B, secs, seed):
	g = torch.Generator().manual_seed(seed)
	x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
	return tuple(t.to(d) for t in (x, torch.linspace(0.6, 1, B), torch.randint(0, 37, (B, 1, 64), generator = g), torch.randint(10, 5 * secs, (B, 1), generator = g)))

shapes = dict(A = (4, 4), B = (3, 5), C = (5, 3))
data = {k: batch(*shapes[k], seed = 30 + i) for i, k in enumerate(sorted(shapes))}

if MODE == 'atencopy':
	from convasr_amd import optimizers as O, functional as Fn, _lib
	def _advance(opt, pair):
		if Fn.capturing():
			_lib.call('convasr_copy', _lib.ptr(pair[1 - opt._cur]), _lib.ptr(pair[opt._cur]), pair[0].numel() * pair.element_size(), _lib.stream_ptr())
		elif getattr(opt, '_pinned', False):
			pair[opt._cur].copy_(pair[1 - opt._cur] * 1.0)  # an ATen kernel instead of hipMemcpyAsync
		else:
			opt._cur = 1 - opt._cur
	O._advance = _advance

KEEP = []


def run(graphed, order, mg):
	ca.functional.manual_seed(23)
	torch.manual_seed(4)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.1, base_width = 64, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = ca.optimizers.AdamW(flat, lr = 1e-3, weight_decay = 1e-2)
	stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed, max_graphs = mg)
	out = []
	for it, k in enumerate(order):
		r = stepper(*data[k], iteration = it)
		out.append((repr(float(r['loss'])), repr(float(r['grad_norm'])), opt.applied.flatten().tolist() if MODE != 'noread' else None, opt._cur))
	torch.cuda.synchronize()
	KEEP.append((model, flat, opt, stepper))  # (the test keeps the eager run's objects alive while the graph run executes)
	return out

e = run(False, 'AABABBCABCA', 64)
g = run(True, 'AABABBCABCA', 2)
for it, (a, b) in enumerate(zip(e, g)):
	print(it, 'AABABBCABCA'[it], 'OK ' if a[:2] == b[:2] else 'DIFF', a, b)
