"""Which side of the eager / replay boundary goes wrong (profiles/r06_interleave_race.txt)?  Parameter, 16-bit mirror and optimizer-state snapshots after
every step of the all-eager run and of the run with graphs (fence off); the first snapshot that differs names the step and the tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import convasr_amd as ca
import test_split_operand_gpu as T

d = torch.device('cuda:0')
shapes = dict(A = (4, 4), B = (3, 5), C = (5, 3))
order = list(sys.argv[1]) if len(sys.argv) > 1 else list('AABABBCABCA')
if os.environ.get('NOFENCE', '1') == '1':
	ca.train.GraphedTrainStep._fence_transition = lambda self, device, eager: None


def run(graphed):
	ca.functional.manual_seed(23)
	torch.manual_seed(4)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.1, base_width = 64, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = ca.optimizers.AdamW(flat, lr = 1e-3, weight_decay = 1e-2)
	stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed, max_graphs = 2)
	data = {k: T._batch(d, *shapes[k], seed = 30 + i) for i, k in enumerate(sorted(shapes))}
	snaps = []
	for it, k in enumerate(order):
		r = stepper(*data[k], iteration = it)
		loss, gn = float(r['loss']), float(r['grad_norm'])
		torch.cuda.synchronize()
		state = {n: v.clone() for n, v in vars(opt).items() if torch.is_tensor(v)}
		snaps.append(dict(loss = loss, gn = gn, params = flat.data.clone(), mirror = None if flat.data16 is None else flat.data16.clone(), grads = flat.grad.clone() if hasattr(flat, 'grad') and torch.is_tensor(flat.grad) else None, state = state, cur = getattr(opt, '_cur', None)))
	return snaps, [n for n, v in vars(opt).items() if torch.is_tensor(v)]


a, names = run(False)
b, _ = run(True)
print('optimizer tensors:', names)
for it, (x, y) in enumerate(zip(a, b)):
	diffs = []
	if x['loss'] != y['loss'] or x['gn'] != y['gn']:
		diffs.append(('loss/gn', x['loss'], y['loss'], x['gn'], y['gn']))
	for key in ('params', 'mirror', 'grads'):
		if x[key] is not None and not torch.equal(x[key], y[key]):
			diffs.append((key, int((x[key] != y[key]).sum()), x[key].numel()))
	for n in x['state']:
		u, v = x['state'][n], y['state'][n]
		if u.shape == v.shape and not torch.equal(u, v):
			diffs.append(('state.' + n, int((u != v).sum()), u.numel(), 'cur', x['cur'], y['cur']))
	print(it, order[it], 'OK' if not diffs else diffs)
