"""Round 6 debug: graphs interleaved with eager steps (AdamW bf16): which state diverges, and when?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import convasr_amd as ca
d = torch.device('cuda:0')

def batch(B, secs, seed):
	g = torch.Generator().manual_seed(seed)
	x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
	return tuple(t.to(d) for t in (x, torch.linspace(0.6, 1, B), torch.randint(0, 37, (B, 1, 64), generator = g), torch.randint(10, 5 * secs, (B, 1), generator = g)))

shapes = dict(A = (4, 4), B = (3, 5), C = (5, 3))
data = {k: batch(*shapes[k], seed = 30 + i) for i, k in enumerate(sorted(shapes))}

def run(graphed, order, max_graphs, optname = 'adamw', dt = torch.bfloat16):
	ca.functional.manual_seed(23)
	torch.manual_seed(4)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.1, base_width = 64, check_time_dim_padded = False, compute_dtype = dt).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = ca.optimizers.AdamW(flat, lr = 1e-3, weight_decay = 1e-2) if optname == 'adamw' else ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed, max_graphs = max_graphs)
	states = []
	for it, k in enumerate(order):
		r = stepper(*data[k], iteration = it)
		torch.cuda.synchronize()
		st = dict(loss = float(r['loss']), params = flat.data.clone(), mirror = None if flat.data16 is None else flat.data16.clone())
		if optname == 'adamw':
			st.update(m = opt.exp_avg.clone(), v = opt.exp_avg_sq.clone(), applied = opt.applied.clone(), cur = opt._cur)
		st['bn'] = torch.cat([b.flatten().float() for n, b in model.named_buffers() if 'running' in n])
		states.append(st)
	return states, stepper

def run_novograd_fp16(graphed, order, mg):
	ca.functional.manual_seed(23)
	torch.manual_seed(4)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.1, base_width = 64, check_time_dim_padded = False, compute_dtype = torch.float16).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
	ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2')
	stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed, max_graphs = mg)
	for it, k in enumerate(order):
		stepper(*data[k], iteration = it)
	torch.cuda.synchronize()

if '--after-novograd' in sys.argv:
	run_novograd_fp16(False, 'AABABBCABCA', 64)
	run_novograd_fp16(True, 'AABABBCABCA', 2)

for optname in ('adamw', 'sgd'):
	for order, mg in (('AABABBCABCA', 2), ('AABABBCABCA', 64), ('AABABBAABBA', 2), ('AACACCA', 1)):
		e, _ = run(False, order, mg, optname)
		g, st = run(True, order, mg, optname)
		print(optname, order, 'max_graphs', mg, 'captures', st.captures, 'replays', st.replays, 'eager', st.eager_steps)
		for it, (a, b) in enumerate(zip(e, g)):
			diffs = {k: (float((a[k].float() - b[k].float()).abs().max()) if torch.is_tensor(a[k]) else (a[k], b[k])) for k in a if k not in ('cur', ) and a[k] is not None}
			bad = {k: v for k, v in diffs.items() if (v != 0.0 if not isinstance(v, tuple) else v[0] != v[1])}
			print('  step', it, order[it], 'loss', a['loss'], b['loss'], 'DIFF' if bad else 'ok', bad if bad else '')
