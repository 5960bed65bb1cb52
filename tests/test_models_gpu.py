"""Module-level parity on the MI355X: the convasr_amd mirror of models.py against golden vectors produced by the reference
(tests/golden/make_golden.py) and against the CPU oracle at larger sizes."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import convasr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
pytestmark = pytest.mark.gpu
T_ = lambda a: torch.as_tensor(np.asarray(a))


def close(a, b, rtol, atol, what = ''):
	a, b = a.detach().double().cpu(), torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().double().cpu()
	assert a.shape == b.shape, (what, a.shape, b.shape)
	err = (a - b).abs()
	tol = atol + rtol * b.abs()
	assert bool((err <= tol).all()), f'{what}: max abs err {float(err.max()):.3e} (max |ref| {float(b.abs().max()):.3e})'


def load_sd(module, g, prefix):
	sd = {k[len(prefix):]: T_(g[k]) for k in g.files if k.startswith(prefix)}
	missing, unexpected = module.load_state_dict(sd, strict = False)
	assert not unexpected, unexpected
	return sd


@pytest.mark.parametrize('ci', range(5))
def test_convbn1d_block_golden(ci):
	import convasr_amd as ca
	case = json.load(open(os.path.join(GOLDEN, 'convblock_cases.json')))[ci]
	g = np.load(os.path.join(GOLDEN, f'convblock{ci}.npz'))
	d = torch.device('cuda:0')
	blk = ca.models.ConvBn1d(num_channels = (case['cin'], case['cout']), kernel_size = case['k'], stride = case['stride'], dilation = case['dilation'], repeat = case['repeat'], nonlinearity = tuple(case['nonlinearity']), temporal_mask = case['temporal_mask'], num_channels_residual = [case['cin']] * case['nres'])
	load_sd(blk, g, f'c{ci}/sd/')
	blk.to(d).train()
	need_gx = case['stride'] == 1
	x = T_(g['x']).to(d).requires_grad_(need_gx)
	frac = T_(g['frac']).to(d)
	res = [T_(g[f'res{r}']).to(d) for r in range(case['nres'])]
	y = blk(x, lengths_fraction = frac, residual = res)
	close(y, g['y'], 1e-4, 1e-4, 'train forward')
	y.backward(T_(g['gout']).to(d))
	if need_gx:
		close(x.grad, g['gx'], 1e-3, 1e-4, 'dx')
	for n, p in blk.named_parameters():
		ref = g[f'c{ci}/grad/{n}']
		close(p.grad, ref, 2e-3, 2e-4 * max(1.0, float(np.abs(ref).max())), 'grad ' + n)
	for k in g.files:
		if k.startswith(f'c{ci}/sd_after/') and 'running' in k:
			close(dict(blk.named_buffers())[k[len(f'c{ci}/sd_after/'):]], g[k], 1e-4, 1e-5, k)
	blk.eval()
	with torch.no_grad():
		close(blk(T_(g['x']).to(d), lengths_fraction = frac, residual = res), g['y_eval'], 1e-4, 1e-4, 'eval forward')
		blk.fuse_conv_bn_eval()
		close(blk(T_(g['x']).to(d), lengths_fraction = frac, residual = res), g['y_eval'], 1e-3, 1e-4, 'fused eval forward')


def _tiny(ca, d, compute_dtype = torch.float32):
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, frontend = fe, check_time_dim_padded = False, nonlinearity = ('hardtanh', 0, 20), dilation = 2, compute_dtype = compute_dtype)
	return model


def test_tiny_model_end_to_end_training_and_decode_golden():
	"""BASELINE config 1 on the GPU: forward, CTC loss, gradients, two SGD steps, eval logits and greedy strings vs the reference."""
	import convasr_amd as ca
	from convasr_amd.transcript_generators import GreedyCTCGenerator, CharTokenizerLegacy
	g = np.load(os.path.join(GOLDEN, 'tiny_e2e.npz'))
	hyp = json.load(open(os.path.join(GOLDEN, 'tiny_e2e_hyp.json')))
	d = torch.device('cuda:0')
	model = _tiny(ca, d)
	sd = load_sd(model, g, 'sd/')
	assert np.array_equal(model.frontend.mel.weight.numpy(), g['sd/frontend.mel.weight'])  # product mel basis == pinned fixture
	model.to(d).train()
	wav, xlen, y, ylen = (T_(g[k]).to(d) for k in ['wav', 'xlen', 'y', 'ylen'])
	flat = ca.train.FlatParameters(model)
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3, keep_clipped_grads = True)
	gen, tok = GreedyCTCGenerator(), CharTokenizerLegacy(O.CHAR_LEGACY_ALPHABET)
	first = lambda tt: [t[0][0]['hyp'] if len(t[0]) else '' for t in tt]

	out = model(wav, xlen, y = y, ylen = ylen)
	close(out['logits'][0], g['step0/logits'], 1e-3, 1e-4, 'logits')  # BASELINE.md fp32 bar: rtol 1e-3 / atol 1e-4
	close(out['log_probs'][0], g['step0/log_probs'], 1e-3, 1e-4, 'log_probs')
	assert torch.equal(out['olen'][0].cpu(), T_(g['step0/olen']))
	close(out['loss'], g['step0/loss_vec'], 1e-4, 0, 'CTC loss (1e-4 relative)')
	assert first(gen.generate(tok, out['log_probs'][0].detach(), torch.zeros(4), torch.ones(4), output_lengths = out['olen'][0])) == hyp['hyp_step0']
	ent = ca.models.entropy(out['log_probs'][0].detach(), out['olen'][0], dim = 1).mean()
	close(ent, g['step0/entropy'], 1e-4, 0, 'entropy')
	loss = (out['loss'] * ylen[:, 0]).mean()
	close(loss, g['step0/loss'], 1e-4, 0, 'loss')
	loss.backward()
	gn = flat.clip_grad_norm_(100.0)
	close(gn, g['step0/grad_norm'], 1e-3, 0, 'grad norm')
	opt.step()
	for k in ['decoder.0.weight', 'backbone.0.conv.0.0.weight', 'backbone.2.bn.0.weight']:
		ref = g['step0/grad/' + k]
		close(dict(model.named_parameters())[k].grad, ref, 5e-3, 1e-3 * float(np.abs(ref).max()), 'clipped grad ' + k)
	opt.zero_grad()

	res = ca.train.train_step(model, opt, wav, xlen, y, ylen)
	close(res['loss'], g['step1/loss'], 1e-3, 0, 'step 1 loss')
	close(res['grad_norm'], g['step1/grad_norm'], 5e-3, 0, 'step 1 grad norm')
	state = model.state_dict()
	assert int(state['backbone.0.bn.0.num_batches_tracked']) == 2
	# (a) against the reference's golden weights.  The reference's own CPU path is only reproducible to ~2e-5 relative in the
	# gradients across thread counts / hosts (measured: grad norm 164.8170 with 1 thread, 164.8184 with 8), and two SGD steps
	# amplify that, so the bar is 5 % of the largest update of each tensor ...
	for k in g.files:
		if k.startswith('sd_after2/') and 'num_batches' not in k:
			name = k[len('sd_after2/'):]
			ref, before = g[k], g['sd/' + name]
			close(state[name], ref, 0, 0.05 * float(np.abs(ref - before).max()) + 1e-6, k)
	# (b) ... and tightly against the oracle taking the same two steps on THIS host's cores.
	osd = {k[3:]: T_(g[k]).clone() for k in g.files if k.startswith('sd/')}
	plan = O.jasper_plan(64, [38], nonlinearity = ('hardtanh', 0, 20), dilation = 2, **O.TINY)
	bufs = {}
	for _ in range(2):
		O.train_step(osd, plan, wav.cpu(), xlen.cpu(), y.cpu(), ylen.cpu(), frontend = dict(nfft = 512, hop_length = 160), momentum_buffers = bufs)
	for name, v in osd.items():
		if v.is_floating_point() and not name.startswith('frontend.'):
			close(state[name], v, 1e-4, 2e-5, 'oracle on this host: ' + name)

	model.eval()
	with torch.no_grad():
		ev = model(wav, xlen)
	close(ev['logits'][0], g['eval_logits'], 0, 3e-2, 'eval logits after two steps (golden; see the reproducibility note above)')
	with torch.no_grad():
		oev = O.jasper_forward(osd, plan, wav.cpu(), xlen.cpu(), frontend = dict(nfft = 512, hop_length = 160), training = False)
	close(ev['logits'][0], oev['logits'], 1e-3, 1e-3, 'eval logits after two steps (oracle on this host)')
	assert first(gen.generate(tok, ev['log_probs'][0], torch.zeros(4), torch.ones(4), output_lengths = ev['olen'][0])) == hyp['hyp']


def test_dense_residual_jaspernet_golden():
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'dense_jasper.npz'))
	d = torch.device('cuda:0')
	model = ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11, 13], out_width_factors = [2, 3], dropouts = [0.2, 0.2], out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, num_subblocks = 2, check_time_dim_padded = False, temporal_mask = False)
	load_sd(model, g, 'sd/')
	model.to(d).train()
	out = model(T_(g['x']).to(d), T_(g['xlen']).to(d))
	close(out['logits'][0], g['logits'], 1e-3, 1e-4, 'dense logits')
	close(out['log_probs'][0], g['log_probs'], 1e-3, 1e-4, 'dense log_probs')
	out['logits'][0].square().mean().backward()
	params = dict(model.named_parameters())
	for k in ['backbone.1.conv_residual.0.weight', 'backbone.0.conv.0.0.weight']:
		ref = g['grad/' + k]
		close(params[k].grad, ref, 5e-3, 5e-3 * float(np.abs(ref).max()), 'grad ' + k)


def test_full_wav2letter_forward_fp32_vs_oracle_config2_shape_reduced_batch():
	"""BASELINE config 2 (full Wav2Letter, fp32, logmel + conv stack + CTC forward) at 4 x 10 s so the CPU oracle finishes in
	seconds; same tolerances as BASELINE.md: logits rtol 1e-3 / atol 1e-4 scaled to the logit range, CTC 1e-4 relative,
	greedy strings identical."""
	import convasr_amd as ca
	from convasr_amd.transcript_generators import GreedyCTCGenerator, CharTokenizerLegacy
	torch.manual_seed(1)
	d = torch.device('cuda:0')
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False)
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	B, secs = 4, 10
	x = torch.rand(B, 16000 * secs) * 2 - 1
	xlen = torch.linspace(0.5, 1, B)
	y = torch.randint(0, 37, (B, 1, 10 * secs))
	ylen = torch.tensor([[50], [60], [80], [100]])
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	with torch.no_grad():
		ref = O.jasper_forward(sd, plan, x, xlen, y, ylen, frontend = dict(nfft = 512, hop_length = 160), training = True)
	model.to(d).train()
	with torch.no_grad():
		out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	scale = float(ref['logits'].abs().max())
	close(out['logits'][0], ref['logits'], 1e-3, 1e-4 * max(scale, 1.0), 'logits')
	close(out['loss'], ref['loss'], 1e-4, 0, 'CTC loss')
	tok, gen = CharTokenizerLegacy(O.CHAR_LEGACY_ALPHABET), GreedyCTCGenerator()
	got = [t[0][0]['hyp'] if len(t[0]) else '' for t in gen.generate(tok, out['log_probs'][0], torch.zeros(B), torch.ones(B), output_lengths = out['olen'][0])]
	want = O.greedy_decode(ref['log_probs'], ref['olen'])
	# a random-init network decides a few of its frames by margins below the difference of two fp32 summation orders: the argmax must
	# agree on every frame the oracle decides by more than twice the observed deviation, and at most one string may differ (the
	# stated-batch version of this test, tests/test_model_variants_gpu.py, has the details)
	lp_dev = float((out['log_probs'][0].cpu() - ref['log_probs']).abs().max())
	top2 = ref['log_probs'].topk(2, dim = 1).values
	decisive = (top2[:, 0] - top2[:, 1]) > 2 * lp_dev
	agree = out['log_probs'][0].argmax(dim = 1).cpu() == ref['log_probs'].argmax(dim = 1)
	assert bool(agree[decisive].all()) and float(decisive.float().mean()) > 0.99 and sum(a == b for a, b in zip(got, want)) >= B - 1, (got, want)
	for k, v in model.state_dict().items():
		if 'running' in k:
			close(v, sd[k], 1e-3, 1e-5, k)  # oracle updated its copy of the running stats in place


def test_bf16_training_step_tracks_fp32():
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'tiny_e2e.npz'))
	d = torch.device('cuda:0')
	losses = {}
	for dt in (torch.float32, torch.bfloat16):
		model = _tiny(ca, d, dt)
		load_sd(model, g, 'sd/')
		model.to(d).train()
		flat = ca.train.FlatParameters(model)
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		wav, xlen, y, ylen = (T_(g[k]).to(d) for k in ['wav', 'xlen', 'y', 'ylen'])
		losses[dt] = [float(ca.train.train_step(model, opt, wav, xlen, y, ylen)['loss']) for _ in range(3)]
	for a, b in zip(losses[torch.float32], losses[torch.bfloat16]):
		assert abs(a - b) / abs(a) < 0.03, losses
	assert losses[torch.bfloat16][2] < losses[torch.bfloat16][0]


# ------------------------------------------------------------------------------------------------ SURVEY 8(f) "next" rows

def test_novograd_matches_reference_golden():
	"""optimizers.NovoGrad after clip_grad_norm_ (tests/golden/make_golden_next.py), 4 steps, two hyper-parameter sets; the
	fused kernel works on the flat arena, so the tensors go through FlatParameters like a model's would."""
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'novograd.npz'))
	d = torch.device('cuda:0')
	for case in (0, 1):
		lr, b1, b2, eps, wd, damp, max_norm = [float(v) for v in g[f'c{case}/hyper']]
		n = len([k for k in g.files if k.startswith(f'c{case}/p0/')])
		holder = torch.nn.Module()
		holder.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(g[f'c{case}/p0/{i}']).to(d)) for i in range(n)])
		flat = ca.train.FlatParameters(holder)
		opt = ca.optimizers.NovoGrad(flat, lr = lr, betas = (b1, b2), eps = eps, weight_decay = wd, dampening = bool(damp))
		for step in range(4):
			for i, p in enumerate(flat.params):
				p._convasr_grad.copy_(torch.from_numpy(g[f'c{case}/g{step}/{i}']).to(d))
				p._convasr_fresh = False
			norm = flat.clip_grad_norm_(max_norm)
			opt.step()
			opt.zero_grad()
			close(norm.cpu(), torch.from_numpy(g[f'c{case}/norm{step}']), 1e-5, 0, 'grad norm')
			close(opt.total_norm.cpu().squeeze(), torch.from_numpy(g[f'c{case}/norm{step}']), 1e-5, 0, 'grad norm (fused)')
			st = opt.state
			for i, p in enumerate(flat.params):
				close(p.detach().cpu(), torch.from_numpy(g[f'c{case}/p{step + 1}/{i}']), 2e-5, 2e-6, f'case {case} step {step} param {i}')
				close(st[p]['_grads_ema'].cpu(), torch.from_numpy(g[f'c{case}/ema{step + 1}/{i}']), 2e-5, 0, f'case {case} step {step} ema {i}')
		# the same steps fed with 4x the gradients and grad_scale = 1/4 (rank-summed gradients, mean folded into the kernel)
		holder2 = torch.nn.Module()
		holder2.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(g[f'c{case}/p0/{i}']).to(d)) for i in range(n)])
		flat2 = ca.train.FlatParameters(holder2)
		opt2 = ca.optimizers.NovoGrad(flat2, lr = lr, betas = (b1, b2), eps = eps, weight_decay = wd, dampening = bool(damp))
		for step in range(4):
			for i, p in enumerate(flat2.params):
				p._convasr_grad.copy_(4 * torch.from_numpy(g[f'c{case}/g{step}/{i}']).to(d))
				p._convasr_fresh = False
			flat2.grad_scale = 0.25
			norm = flat2.clip_grad_norm_(max_norm)
			opt2.step()
			opt2.zero_grad()
			close(norm.cpu(), torch.from_numpy(g[f'c{case}/norm{step}']), 1e-5, 0, 'grad norm (scaled)')
		for i, p in enumerate(flat2.params):
			close(p.detach().cpu(), torch.from_numpy(g[f'c{case}/p4/{i}']), 2e-5, 2e-6, f'case {case} grad_scale param {i}')
		# a gated (non-finite loss) step changes nothing, including the EMAs
		before, ema = flat.data.clone(), opt.state[flat.params[0]]['_grads_ema'].clone()
		for p in flat.params:
			p._convasr_fresh = False
		flat.clip_grad_norm_(max_norm)
		opt.step(loss_gate = torch.tensor([float('nan')], device = d))
		assert torch.equal(flat.data, before) and torch.equal(opt.state[flat.params[0]]['_grads_ema'], ema)


def test_novograd_train_step_on_tiny_model_tracks_oracle():
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.manual_seed(0)
	model = ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.0], out_width_factors_large = [2, 2], residual = False, repeat = 1, check_time_dim_padded = False, temporal_mask = False).to(d).train()
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.98), weight_decay = 1e-3)
	ref_params = [p.detach().cpu().clone() for p in flat.params]
	state = {}
	x, xlen = torch.randn(2, 64, 40, device = d), torch.ones(2, device = d)
	y, ylen = torch.randint(0, 37, (2, 1, 6), device = d), torch.tensor([[5], [4]], device = d)
	for it in range(3):
		grads_before = None
		res = ca.train.train_step(model, opt, x, xlen, y, ylen, max_norm = 1.0)
		assert not bool(res['skipped'])
		# replay the same update with the oracle's NovoGrad on the gradients the device produced
		grads = [p._convasr_grad.detach().cpu().clone() for p in flat.params]
		O.novograd_step(ref_params, grads, state, lr = 1e-3, betas = (0.95, 0.98), eps = 1e-8, weight_decay = 1e-3, dampening = False, max_norm = 1.0)
		for p, r in zip(flat.params, ref_params):
			close(p.detach().cpu(), r, 1e-4, 1e-6, f'iteration {it}')


def test_ctc_alignment_matches_reference_golden():
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'alignment.npz'))
	d = torch.device('cuda:0')
	for case in (0, 1, 2):
		T_ = lambda k: torch.from_numpy(g[f'c{case}/{k}'])
		al = ca.ctc.alignment(T_('log_probs').to(d), T_('targets'), T_('input_lengths'), T_('target_lengths'), blank = int(g[f'c{case}/blank']))
		assert torch.equal(al.cpu(), T_('alignment')), (case, (al.cpu() != T_('alignment')).sum())


def test_ctc_alignment_properties_at_benchmark_size():
	"""64 x 753 frames x 150 labels: the alignment of every utterance is strictly increasing over its labels, inside
	[0, input_length), padded labels are 0, and it equals the oracle's on a sample of utterances."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	B, C, T, S = 64, 38, 753, 150
	gen = torch.Generator().manual_seed(5)
	lp = torch.randn(T, B, C, generator = gen).log_softmax(dim = -1)
	tg = torch.randint(0, C - 1, (B, S), generator = gen)
	il = torch.randint(2 * S + 1, T + 1, (B, ), generator = gen)
	tl = torch.randint(S // 2, S + 1, (B, ), generator = gen)
	al = ca.ctc.alignment(lp.to(d), tg, il, tl, blank = C - 1).cpu()
	for b in range(B):
		a = al[b, :tl[b]]
		assert bool((a[1:] > a[:-1]).all()) and int(a[0]) >= 0 and int(a[-1]) < int(il[b])
		assert int(al[b, tl[b]:].abs().sum()) == 0
	sample = [0, 17, 63]
	ref = O.ctc_alignment(lp[:, sample], tg[sample], il[sample], tl[sample], blank = C - 1)
	assert torch.equal(al[sample], ref)


def test_fused_eval_forward_under_hip_graph_replay_is_bit_identical():
	"""f1: the fused eval forward (frontend -> instance norm -> BN-folded conv stack -> decoder -> log-softmax) captured into a
	HIP graph replays to the same bits as the eager launches, for new input written into the static buffer."""
	import bench_infer
	d = torch.device('cuda:0')
	torch.manual_seed(3)
	model = bench_infer.build_model('Wav2Letter', d, torch.bfloat16)
	xlen = torch.ones(1, device = d)
	static_in = torch.empty(1, 48000, device = d)
	with torch.no_grad():
		static_in.copy_(torch.rand(1, 48000) * 2 - 1)
		side = torch.cuda.Stream()
		side.wait_stream(torch.cuda.current_stream())
		with torch.cuda.stream(side):
			for _ in range(2):
				model(static_in, xlen)
		torch.cuda.current_stream().wait_stream(side)
		torch.cuda.synchronize()
		graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(graph):
			out = model(static_in, xlen)
		for seed in (10, 11):
			x = torch.rand(1, 48000, generator = torch.Generator().manual_seed(seed)) * 2 - 1
			static_in.copy_(x)
			graph.replay()
			torch.cuda.synchronize()
			replayed = out.clone()
			eager = model(x.to(d), xlen)
			assert torch.isfinite(eager).all() and torch.equal(replayed, eager)


def test_cross_layer_backward_fusion_matches_separate_reduce():
	"""bf16 training: pass 1 of the batch-norm backward fused into the producing dgrad launch's epilogue (functional._dgrad) vs
	the separate reduce kernel: same dz bits go into both, so the parameter gradients agree to fp32 summation-order noise."""
	import convasr_amd as ca
	from convasr_amd import functional as Fn
	d = torch.device('cuda:0')
	grads, fused_calls = {}, {}
	for fuse in (True, False):
		Fn.FUSE_BWD = fuse
		try:
			torch.manual_seed(0)
			ca.functional.manual_seed(5)
			model = ca.models.JasperNet(64, [38], base_width = 64, kernel_sizes = [11, 13], out_width_factors = [2, 3], dropouts = [0.2, 0.2], out_width_factors_large = [4, 4], residual = False, repeat = 2, dropout = 0.2, check_time_dim_padded = False, temporal_mask = True, nonlinearity = ('hardtanh', 0, 20), compute_dtype = torch.bfloat16).to(d).train()
			flat = ca.train.FlatParameters(model)
			model._convasr_flat = flat
			g = torch.Generator().manual_seed(1)
			x = torch.randn(3, 64, 300, generator = g).to(d)
			xlen = torch.tensor([1.0, 0.7, 0.45], device = d)
			y = torch.randint(0, 37, (3, 1, 20), generator = g).to(d)
			ylen = torch.tensor([[20], [15], [9]], device = d)
			calls = []
			orig = ca.ops.conv1d_dgrad_bn_reduce
			ca.ops.conv1d_dgrad_bn_reduce = lambda *a, **k: (lambda r: (calls.append(r is not None), r)[1])(orig(*a, **k))
			try:
				out = model(x, xlen, y = y, ylen = ylen)
				(out['loss'] * ylen[:, 0]).mean().backward()
			finally:
				ca.ops.conv1d_dgrad_bn_reduce = orig
			flat.finalize_grads()
			torch.cuda.synchronize()
			grads[fuse] = flat.grad.clone()
			fused_calls[fuse] = calls
		finally:
			Fn.FUSE_BWD = True
	assert sum(fused_calls[True]) >= 5 and not fused_calls[False], fused_calls
	a, b = grads[True], grads[False]
	assert torch.isfinite(a).all() and float(b.abs().max()) > 0
	# the two paths differ only in the fp32/fp64 summation order of the per-channel sums; in a bf16 network that moves a few
	# activations' gradients by one bf16 ulp downstream, so the comparison is in norm, not element by element
	rel = float((a - b).norm() / b.norm())
	# (with the channel-padded decoder backward the dgrad of the head carries the last layer's sums too: one more fused layer, 2.4e-3)
	assert rel < 4e-3 and float((a - b).abs().max()) < 4e-3 * float(b.abs().max()), (rel, float((a - b).abs().max()), float(b.abs().max()))


def test_training_is_bitwise_reproducible():
	"""No floating-point atomics on the SGD training path: two runs of the same bf16 program (dropout on, masks on, fused
	backward epilogue on) end with bit-identical parameters and report the same loss at every step."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	finals, losses = [], []
	for run in range(2):
		torch.manual_seed(0)
		ca.functional.manual_seed(7)
		fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
		model = ca.models.JasperNet(64, [38], base_width = 64, kernel_sizes = [11, 13], out_width_factors = [2, 3], dropouts = [0.2, 0.2], out_width_factors_large = [4, 4], residual = False, repeat = 2, dropout = 0.2, frontend = fe, check_time_dim_padded = False, nonlinearity = ('hardtanh', 0, 20), compute_dtype = torch.bfloat16).to(d).train()
		flat = ca.train.FlatParameters(model)
		model._convasr_flat = flat
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		g = torch.Generator().manual_seed(3)
		x = (torch.rand(6, 16000 * 3, generator = g) * 2 - 1).to(d)
		xlen = torch.tensor([1.0, 0.9, 0.55, 0.7, 1.0, 0.8], device = d)
		y = torch.randint(0, 37, (6, 1, 25), generator = g).to(d)
		ylen = torch.randint(10, 26, (6, 1), generator = g).to(d)
		ls = []
		for it in range(4):
			r = ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = it)
			ls.append((float(r['loss_cur']), float(r['grad_norm'])))
		finals.append(flat.data.clone())
		losses.append(ls)
	assert losses[0] == losses[1], losses
	assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize('name', ['f32', 'bf16'])
def test_one_request_alone_equals_its_row_of_a_full_batch(name):
	"""Online inference cuts a launch of a few tiles over the input channels (convasr_conv1d_fwd_splitk: every conv of a 1 x 3 s request), a batch
	that fills the chip runs the unsplit kernels: the same utterance gives the same log-probs either way -- exact fp32: 1e-5 absolute and the same
	greedy path; bf16 storage: within its rounding -- and with the split switched off the request reproduces itself to the same bars."""
	import bench_infer
	from convasr_amd import ops
	d = torch.device('cuda:0')
	torch.manual_seed(3)
	dt = dict(f32 = torch.float32, bf16 = torch.bfloat16)[name]
	model = bench_infer.build_model('Wav2Letter', d, dt)
	x = (torch.rand(48, 48000, generator = torch.Generator().manual_seed(4)) * 2 - 1).to(d)
	calls = []
	call = ops.call
	ops.call = lambda fn, *a: (calls.append(fn), call(fn, *a))[1]
	try:
		with torch.no_grad():
			full = model(x, torch.ones(48, device = d))
			n_full = calls.count('convasr_conv1d_fwd_splitk')
			solo = model(x[:1], torch.ones(1, device = d))
			n_solo = calls.count('convasr_conv1d_fwd_splitk') - n_full
			ops.SPLITK = False
			unsplit = model(x[:1], torch.ones(1, device = d))
	finally:
		ops.call, ops.SPLITK = call, True
	assert n_full == 0 and n_solo >= 17, (n_full, n_solo)  # every conv of the request but the 64-channel prologue and what is too small to cut
	bar = 1e-5 if name == 'f32' else 6e-2
	for other in (full[:1], unsplit):
		assert float((solo.float() - other.float()).abs().max()) <= bar, float((solo.float() - other.float()).abs().max())
	if name == 'f32':
		assert torch.equal(solo.argmax(1), full[:1].argmax(1))
