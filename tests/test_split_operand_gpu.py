"""The split-operand ("x3") conv path on the MI355X (csrc/split3.hip, functional.split_weight, JasperNet.set_compute_dtype('bf16x3')):
fp32 storage, every stride-1 conv as hi*hi + hi*lo + lo*hi on the 16-bit matrix pipe.  Reference arithmetic: nn.Conv1d in fp32
(models.py:47-77); bars: the planes are exact splits, the three conv directions agree with float64 to the split's own 2^-16 / 2^-22, a
training step agrees with the exact-fp32 path to 1e-5 in the CTC loss (north_star: 1e-4 against the reference; the 64 x 15 s case against
the CPU oracle lives in test_full_size_and_step_graphs_gpu.py), and replays bit for bit from a step graph."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
	a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
	return float((a - b).norm() / b.norm())


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_split3_planes_are_the_exact_hi_lo_split_in_both_orders(dt):
	from convasr_amd import ops
	d = torch.device('cuda:0')
	torch.manual_seed(0)
	B, C, T = 3, 72, 41
	x = ops.as_cl((torch.randn(B, C, T) * torch.logspace(-3, 1, C).view(1, C, 1)).to(d), torch.float32)
	hi = x.to(dt)
	lo = (x - hi.float()).to(dt)
	for order, planes in ((ops.SPLIT_INPUT, (hi, lo, hi)), (ops.SPLIT_GRAD, (hi, hi, lo))):
		x3 = ops.split3(x, dt, order)
		assert x3.shape == (B, 3 * C, T) and ops.is_cl(x3)
		for p, want in enumerate(planes):
			assert torch.equal(x3[:, p * C:(p + 1) * C, :], want), (order, p)
		# the weight gradient's view of the same memory: frame 3 t + p = plane p of frame t
		xf = ops.split3_frames(x3)
		assert xf.shape == (B, C, 3 * T) and ops.is_cl(xf) and xf.data_ptr() == x3.data_ptr()
		for p, want in enumerate(planes):
			assert torch.equal(xf[:, :, p::3], want), (order, p)
	bits = 8 if dt == torch.bfloat16 else 11
	# what the two planes leave out: 2^-16 (bf16) / 2^-22 (fp16) of the value -- or, for fp16, half a step of its subnormal grid (2^-25), whichever is larger
	resid = (hi.float() + lo.float() - x).abs()
	assert bool((resid <= torch.maximum(x.abs() * 2.0 ** (-2 * bits), torch.full_like(x, 2.0 ** -25 if dt == torch.float16 else 0.0))).all())


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('kmajor', [False, True])
def test_pack_conv_weight_split3_layouts(dt, kmajor):
	from convasr_amd import ops
	d = torch.device('cuda:0')
	torch.manual_seed(1)
	Cout, Cin, K = 136, 192, 5
	w = (torch.randn(Cout, Cin, K) / 30).to(d)
	if kmajor:  # the training arena's element order [K][Cout][Cin] behind the same (Cout, Cin, K) view
		w = w.permute(2, 0, 1).contiguous().permute(1, 2, 0)
		assert ops.weight_layout(w) == 1
	fwd, dgr = ops.pack_weight_split3(w, dt)
	hi = w.to(dt)
	lo = (w - hi.float()).to(dt)
	assert fwd.shape == (K, ops.cout_pad(Cout), 3 * Cin) and dgr.shape == (K, ops.cout_pad(Cin), 3 * Cout)
	for k in range(K):
		for p, want in enumerate((hi, hi, lo)):
			assert torch.equal(fwd[k, :Cout, p * Cin:(p + 1) * Cin], want[:, :, k]), ('fwd', k, p)
		for p, want in enumerate((hi, lo, hi)):
			assert torch.equal(dgr[K - 1 - k, :Cin, p * Cout:(p + 1) * Cout], want[:, :, k].t()), ('dgrad', k, p)
	assert float(fwd[:, Cout:].abs().max()) == 0 and float(dgr[:, Cin:].abs().max()) == 0
	# refreshed in place: same buffers
	f2, d2 = ops.pack_weight_split3(w * 2, dt, out = (fwd, dgr))
	assert f2.data_ptr() == fwd.data_ptr() and d2.data_ptr() == dgr.data_ptr() and torch.equal(f2[0, :Cout, :Cin], (w * 2).to(dt)[:, :, 0])


@pytest.mark.parametrize('dt,tol', [(torch.bfloat16, 1.5e-5), (torch.float16, 6e-5)])
@pytest.mark.parametrize('shape', [(3, 256, 384, 300, 11, 1, 5), (2, 768, 896, 200, 29, 2, 29), (3, 896, 1024, 257, 1, 1, 0), (2, 192, 136, 77, 3, 1, 1)])
def test_split_conv_forward_dgrad_wgrad_against_float64(dt, tol, shape):
	"""(fp16 planes: the test's output gradient of ~1e-3 puts its lo plane into fp16's subnormals -- 1.7e-5; a training run scales the loss)"""
	from convasr_amd import ops, functional as Fn
	d = torch.device('cuda:0')
	B, Cin, Cout, T, K, dil, pad = shape
	torch.manual_seed(2)
	x = torch.randn(B, Cin, T)
	w = torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
	ref = torch.nn.functional.conv1d(x.double(), w.double(), padding = pad, dilation = dil)
	dy = torch.randn_like(ref).float() * 1e-3
	dx_ref = torch.nn.grad.conv1d_input(x.shape, w.double(), dy.double(), padding = pad, dilation = dil)
	dw_ref = torch.nn.grad.conv1d_weight(x.double(), w.shape, dy.double(), padding = pad, dilation = dil)
	xg, wg, dyg = ops.as_cl(x.to(d), torch.float32), w.to(d), ops.as_cl(dy.to(d), torch.float32)
	x3, dy3 = ops.split3(xg, dt, ops.SPLIT_INPUT), ops.split3(dyg, dt, ops.SPLIT_GRAD)
	wf, wd = Fn.split_weight(wg, dt)
	y = ops.conv1d(x3, wf, Cout, K, 1, dil, pad, out_dtype = torch.float32)
	dx = ops.conv1d(dy3, wd, Cin, K, 1, dil, dil * (K - 1) - pad, out_dtype = torch.float32)
	dw = torch.empty(Cout, Cin, K, device = d)
	ops.conv1d_wgrad(ops.split3_frames(x3), ops.split3_frames(dy3), Cout, K, 1, 3 * dil, 3 * pad, dw)
	errs = dict(y = _rel(y, ref), dx = _rel(dx, dx_ref), dw = _rel(dw, dw_ref))
	assert y.dtype == dx.dtype == torch.float32 and max(errs.values()) <= tol, errs
	# against the oracle's restatement of the SAME arithmetic (three kept products of the planes, summed in float64): what separates the kernel from it is
	# its fp32 accumulation of 3 Cin K products only (measured 7.5e-7 at 256 x 11, 2.2e-6 at 768 x 29: below what the split itself leaves out, 4.5e-6)
	from oracle import convasr_oracle as O
	assert _rel(y, O.conv1d_split3(x, w, padding = pad, dilation = dil, dtype = dt)) <= 3e-6


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_bn_passes_write_the_planes_the_separate_split_pass_would(dt):
	"""convasr_bn_act_fwd_split3 / convasr_bn_act_bwd_apply_split3: the fp32 pass's values, split exactly as convasr_split3 splits them
	(dropout, gates, length masks, a residual input): bit-identical planes and gate bits."""
	from convasr_amd import ops, _lib
	d = torch.device('cuda:0')
	torch.manual_seed(7)
	B, C, T = 3, 136, 53
	y = ops.as_cl(torch.randn(B, C, T).to(d) * 3, torch.float32)
	r = ops.as_cl(torch.randn(B, C, T).to(d), torch.float32)
	scale, shift = (torch.rand(C) + 0.5).to(d), torch.randn(C).to(d)
	xl = torch.tensor([1.0, 0.7, 0.4], device = d)
	act = (_lib.ACT_HARDTANH, 0.0, 20.0)
	for res in ((), (r, )):
		kw = dict(xlen = xl, res = list(res), rscale = [None] * len(res), rshift = [None] * len(res), dropout_p = 0.2, seed = 5, offset = 11)
		g1, g2 = (torch.zeros(B * T * C // 8, dtype = torch.uint8, device = d) for _ in range(2))
		z = ops.bn_act(y, scale, shift, act, gate = g1, **kw)
		z3 = ops.bn_act(y, scale, shift, act, gate = g2, planes = dt, **kw)
		assert torch.equal(z3, ops.split3(z, dt, ops.SPLIT_INPUT)) and torch.equal(g1, g2)
	coef = torch.randn(3 * C, device = d)
	dz = ops.as_cl(torch.randn(B, C, T).to(d) * 1e-2, torch.float32)
	for gate in (None, g1):
		kw = dict(xlen = xl, dropout_p = 0.2, seed = 5, offset = 11, gate = gate)
		dy = ops.bn_act_bwd_apply(dz, y, coef, True, scale, shift, act, **kw)
		dy3 = ops.bn_act_bwd_apply(dz, y, coef, True, scale, shift, act, planes = dt, **kw)
		assert torch.equal(dy3, ops.split3(dy, dt, ops.SPLIT_GRAD))


def test_planes_only_outputs_are_refused_by_every_other_reader():
	"""A layer whose one reader is a split conv hands autograd a placeholder of the output's logical shape (functional._planes_placeholder):
	NaN behind stride 0, the planes hanging on it.  ops.as_cl -- the door every consumer of this package goes through -- refuses it."""
	import convasr_amd as ca
	from convasr_amd import ops, functional as Fn
	d = torch.device('cuda:0')
	model = _small(ca, d, 'bf16x3')
	x, xlen, y, ylen = _batch(d, 3, 3)
	feats = model.normalize_features(model.frontend(x, xlen = xlen), xlen = xlen, out_dtype = torch.float32)
	z = model.backbone[0](feats, lengths_fraction = xlen)
	assert z.shape[1] == 128 and z.dtype == torch.float32 and z.stride() == (0, 0, 0) and bool(torch.isnan(z).all())
	with pytest.raises(ca._lib.ConvasrHipError):
		ops.as_cl(z, torch.float32)
	planes = z.__dict__[Fn._PLANES_ATTR]
	assert planes.shape == (3, 3 * 128, z.shape[2]) and planes.dtype == torch.bfloat16
	out = model.backbone[1](z, lengths_fraction = xlen)  # the wired reader takes the planes
	assert Fn._PLANES_ATTR not in z.__dict__ and out.shape[1] == 128 and Fn._PLANES_ATTR in out.__dict__  # (block 1 feeds block 2 alone: a placeholder again)
	os.environ['CONVASR_NO_PLANES_OUT'] = '1'
	try:
		z2 = model.backbone[0](feats, lengths_fraction = xlen)
		assert ops.is_cl(z2) and bool(torch.isfinite(z2).all()) and torch.equal(ops.split3(z2, torch.bfloat16, ops.SPLIT_INPUT), planes)
	finally:
		del os.environ['CONVASR_NO_PLANES_OUT']


def _small(ca, d, dt, dropout = 0.0, seed = 3):
	torch.manual_seed(seed)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	return ca.models.Wav2Letter(64, [38], frontend = fe, dropout = dropout, base_width = 64, check_time_dim_padded = False, compute_dtype = dt).to(d).train()


def _batch(d, B, secs, seed = 5):
	g = torch.Generator().manual_seed(seed)
	x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
	xlen = torch.linspace(0.6, 1, B)
	y = torch.randint(0, 37, (B, 1, 64), generator = g)
	ylen = torch.randint(10, 5 * secs, (B, 1), generator = g)
	return tuple(t.to(d) for t in (x, xlen, y, ylen))


@pytest.mark.parametrize('name', ['bf16x3', 'f16x3'])
def test_split_operand_training_step_tracks_the_exact_fp32_path(name):
	"""Two SGD steps of a small Wav2Letter: the first loss within 1e-5 of the exact-fp32 path's, parameters after the steps within 20 % of the
	largest update (a sanity bound); the split convs are the launches that ran (16-bit operands in, fp32 out)."""
	import convasr_amd as ca
	from convasr_amd import _lib
	d = torch.device('cuda:0')
	batch = _batch(d, 6, 4)
	res = {}
	for dt in (torch.float32, name):
		model = _small(ca, d, dt)
		assert (model.split_dtype is not None) == (dt == name) and model.compute_dtype == torch.float32
		flat = ca.train.FlatParameters(model)
		p0 = flat.data.clone()
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		timer = _lib.KernelTimer(only = ())
		_lib.timer = timer
		try:
			losses = [float(ca.train.train_step(model, opt, *batch)['loss']) for _ in range(2)]
		finally:
			_lib.timer = None
		torch.cuda.synchronize()
		res[dt] = (losses, flat.data.clone(), p0, [f for f, _ in timer.sequence])
	fams = res[name][3]
	# per step: 17 stride-1 layers forward + dgrad, the folded prologue forward, the head forward + dgrad; a weight gradient for each of the 19
	assert fams.count('conv1d_igemm_v2s_kernel<x3>') == 2 * (2 * 17 + 1 + 2) and fams.count('conv1d_wgrad<x3>') == 2 * 19 and 'conv1d_igemm_v2s_kernel<x3>' not in res[torch.float32][3]
	assert not any(f in ('conv1d_igemm (other variants)', 'conv1d_wgrad') for f in fams)  # no exact-fp32 conv launch is left in the split step
	assert fams.count('hbm:split3_kernel') == 2 * 3  # per step: the folded prologue's input, the head's input and the head's output gradient; every other plane tensor is written by the pass that produces the values (bn_act forward / backward apply)
	# (first step: the same parameters in both runs; second step: after one update each -- two exact-fp32 implementations of this random-init
	# network already differ by ~1e-2 in their deep gradients, summation-order noise amplified per layer: DESIGN section 2)
	for (a, b), bar in zip(zip(res[name][0], res[torch.float32][0]), (1e-5, 5e-3)):  # (measured: 6e-8 / 1.0e-3 with bf16 planes -- the second loss sits behind an lr = 1e-2 step that took the loss from 482 to 353)
		assert abs(a - b) / abs(b) <= bar, (res[name][0], res[torch.float32][0])
	step = float((res[torch.float32][1] - res[torch.float32][2]).abs().max())
	assert float((res[name][1] - res[torch.float32][1]).abs().max()) <= 0.2 * step, (float((res[name][1] - res[torch.float32][1]).abs().max()), step)  # (measured 9 % of the largest update after the second step; the first step's gradients are held to the oracle at full size in test_full_size_and_step_graphs_gpu.py)


def test_split_operand_dense_residual_network_tracks_the_exact_fp32_path():
	"""A dense-residual JasperNet (1x1 residual branches with their own batch norms, relu, dropout off) in 'bf16x3': the main convs run as split
	convs -- their inputs arrive as planes written by the residual form of the activation pass, their output gradients are split from the fp32
	result of the BN backward -- and so do the one-tap branches (the tapped block output's planes are made once and shared by its readers).  Loss within 1e-5 and every gradient within
	2e-2 relative L2 of the exact-fp32 path's (same parameters)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	x, xlen, y, ylen = _batch(d, 4, 4)
	out = {}
	for dt in (torch.float32, 'bf16x3'):
		torch.manual_seed(4)
		fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
		model = ca.models.JasperNet(64, [38], frontend = fe, base_width = 64, kernel_sizes = [11, 13, 17], out_width_factors = [2, 3, 4], dropouts = [0.0] * 3, out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, num_subblocks = 2, dropout = 0, check_time_dim_padded = False, temporal_mask = False, compute_dtype = dt).to(d).train()
		flat = ca.train.FlatParameters(model)
		timer = ca._lib.KernelTimer(only = ())
		ca._lib.timer = timer
		try:
			res = model(x, xlen, y = y, ylen = ylen)
			(res['loss'] * ylen[:, 0]).mean().backward()
		finally:
			ca._lib.timer = None
		ca.functional.join_side_streams()
		flat.finalize_grads()
		torch.cuda.synchronize()
		out[dt] = (res['loss'].detach().clone(), {n: p._convasr_grad.clone() for n, p in model.named_parameters() if hasattr(p, '_convasr_grad')}, [f for f, _ in timer.sequence])
	a, b = out[torch.float32], out['bf16x3']
	assert not any(f in ('conv1d_igemm (other variants)', 'conv1d_wgrad') for f in b[2]) and 'conv1d_igemm_v2s_kernel<x3>' in b[2]  # the one-tap residual branches run as split convs too: no exact-fp32 conv launch is left
	assert float(((a[0] - b[0]).abs() / a[0].abs()).max()) <= 1e-5
	worst = max((_rel(b[1][n], a[1][n]), n) for n in a[1] if float(a[1][n].abs().max()) > 0)
	assert worst[0] <= 2e-2, worst  # (measured 5.6e-3, in the prologue's batch-norm bias: the far end of a backward pass that amplifies any difference ~1.2x per layer)


def test_f16x3_trains_under_the_dynamic_loss_scaler():
	"""fp16 planes carry 22 significant bits but fp16's range: data_parallel_and_autocast(compute_dtype = 'f16x3') attaches the dynamic loss scaler
	(the output gradients' planes are what over- / underflows).  From a deliberately high initial scale the start-up steps overflow and are skipped
	(parameters untouched), the scale halves until the gradients fit, and the first applied step's loss equals the exact-fp32 path's to 1e-5."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	batch = _batch(d, 6, 4)
	model = _small(ca, d, torch.float32)
	flat = ca.train.FlatParameters(model)
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	ca.models.data_parallel_and_autocast(model, opt, compute_dtype = 'f16x3')
	assert model.split_dtype == torch.float16 and model.compute_dtype == torch.float32 and flat.loss_scaler is not None
	flat.loss_scaler = ca.train.LossScaler(d, init_scale = 2.0 ** 24)
	p0 = flat.data.clone()
	ref = _small(ca, d, torch.float32)
	ref_loss = float(ca.train.train_step(ref, ca.train.SGD(ca.train.FlatParameters(ref), lr = 1e-2, momentum = 0.9, weight_decay = 1e-3), *batch)['loss'])
	hist = []
	for it in range(12):
		untouched = bool(torch.equal(flat.data, p0))
		r = ca.train.train_step(model, opt, *batch, iteration = it)
		hist.append((float(r['loss']), bool(torch.isfinite(r['grad_norm'])), flat.loss_scaler.loss_scale(), untouched))
	skipped = [h for h in hist if not h[1]]
	assert skipped and all(h[3] for h in hist[:len(skipped) + 1]) and any(h[1] for h in hist), hist  # overflowed steps first, parameters untouched until the first applied one
	assert hist[-1][2] < 2.0 ** 24 and abs(hist[0][0] - ref_loss) / ref_loss <= 1e-5, (hist, ref_loss)
	assert bool(torch.isfinite(flat.data).all()) and not torch.equal(flat.data, p0)


def test_split_operand_eval_and_no_grad_run_the_exact_fp32_kernels():
	import convasr_amd as ca
	d = torch.device('cuda:0')
	x, xlen, y, ylen = _batch(d, 4, 3)
	a, b = _small(ca, d, torch.float32), _small(ca, d, 'bf16x3')
	with torch.no_grad():
		la, lb = a(x, xlen)['logits'][0], b(x, xlen)['logits'][0]
	assert torch.equal(la, lb)
	a.eval(); b.eval()
	with torch.no_grad():
		assert torch.equal(a(x, xlen)['logits'][0], b(x, xlen)['logits'][0])


@pytest.mark.parametrize('name', ['bf16x3', 'f16x3'])
def test_split_operand_inference_is_opt_in_and_tracks_the_exact_fp32_logits(name):
	"""set_compute_dtype(name, inference = True): the fused evaluation path (fuse_conv_bn_eval: bias + activation + mask in the conv epilogue) and
	a dense-residual network's eval-mode 1x1 branches as split convs -- logits within 2e-4 of their range of the exact-fp32 path's, greedy
	paths equal wherever fp32 decides by a margin."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	x, xlen, y, ylen = _batch(d, 4, 4)
	for build in (lambda: _small(ca, d, torch.float32), lambda: ca.models.JasperNet(64, [38], frontend = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window'), base_width = 64, kernel_sizes = [11, 13], out_width_factors = [2, 3], dropouts = [0.0, 0.0], out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, check_time_dim_padded = False, temporal_mask = False).to(d).train()):
		torch.manual_seed(11)
		model = build()
		with torch.no_grad():
			model(x, xlen)  # one train-mode pass: running statistics that are not the initial ones
		model.eval()
		model.fuse_conv_bn_eval()
		with torch.no_grad():
			ref = model(x, xlen)['log_probs'][0].clone()
			model.set_compute_dtype(name, inference = True)
			assert model.backbone[1].split_inference
			got = model(x, xlen)['log_probs'][0]
			model.set_compute_dtype(name)
			assert not model.backbone[1].split_inference and torch.equal(model(x, xlen)['log_probs'][0], ref)  # (default: evaluation stays exact fp32)
		rng = float(ref.max() - ref.min())
		assert float((got - ref).abs().max()) <= 2e-4 * rng, (float((got - ref).abs().max()), rng)
		top2 = ref.topk(2, dim = 1).values
		decided = (top2[:, 0] - top2[:, 1]) > 1e-3 * rng
		assert bool((got.argmax(1) == ref.argmax(1))[decided].all())


def test_split_operand_step_replays_bitwise_from_a_graph_with_dropout():
	import convasr_amd as ca
	d = torch.device('cuda:0')
	batches = [_batch(d, 5, 4, seed = 10 + i) for i in range(8)]
	out = {}
	for graphed in (False, True):
		ca.functional.manual_seed(17)
		model = _small(ca, d, 'bf16x3', dropout = 0.2)
		flat = ca.train.FlatParameters(model)
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed)
		trace = [float(stepper(*b, iteration = i)['loss']) for i, b in enumerate(batches)]
		torch.cuda.synchronize()
		out[graphed] = (trace, flat.data.clone(), stepper)
	assert out[True][2].captures == 1 and out[True][2].replays == 7
	assert out[False][0] == out[True][0] and torch.equal(out[False][1], out[True][1])


# ------------------------------------------------------------------------------------------------ graphs and eager steps interleaved

def _interleaved(ca, d, make_opt, dt, opt_level, graphed, order, shapes, max_graphs = 64, validate_before_capture = False):
	ca.functional.manual_seed(23)
	torch.manual_seed(4)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.1, base_width = 64, check_time_dim_padded = False, compute_dtype = dt).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = make_opt(flat)
	if opt_level is not None:
		ca.models.data_parallel_and_autocast(model, opt, opt_level = opt_level)
	stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed, max_graphs = max_graphs)
	data = {k: _batch(d, *shapes[k], seed = 30 + i) for i, k in enumerate(sorted(shapes))}
	trace = []
	for it, k in enumerate(order):
		if validate_before_capture and it in validate_before_capture:
			model.eval()
			with torch.no_grad():
				model(data[k][0], data[k][1])  # a validation pass between two optimizer steps: refreshes the version-keyed packed copies
			model.train()
		r = stepper(*data[k], iteration = it)
		trace.append((repr(float(r['loss'])), repr(float(r['grad_norm']))))  # (repr: the NaN norm of an overflowed fp16 start-up step must compare equal to itself)
	torch.cuda.synchronize()
	scaler = None if flat.loss_scaler is None else flat.loss_scaler.current.tolist()
	return trace, flat.data.clone(), scaler, stepper


@pytest.mark.parametrize('optname', ['novograd_fp16', 'adamw_bf16'])
def test_graphs_interleaved_with_eager_steps_keep_the_double_buffered_state_current(optname):
	"""Shape order A A B A B B C A B: the eager warm-up of B (and of C, which stays eager for good with max_graphs = 2) runs AFTER graph A
	was captured.  NovoGrad's EMAs / AdamW's applied-step counter / the fp16 loss scaler are double-buffered on the device; a flip by those
	eager steps would leave graph A reading a row that is one step old.  Bit for bit against the eager run -- with NO host wait between eager and
	replayed steps: every captured step consists of kernel nodes only, so GraphedTrainStep's transition fence is not armed (a memset node recorded for
	convasr_signal_absmax's hipMemsetAsync made this very sequence differ at step 10 in most runs: profiles/r06_interleave_race.txt)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	shapes = dict(A = (4, 4), B = (3, 5), C = (5, 3))
	order = list('AABABBCABCA')
	if optname == 'novograd_fp16':
		make_opt, dt, lvl = (lambda flat: ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)), torch.float16, 'O2'
	else:
		make_opt, dt, lvl = (lambda flat: ca.optimizers.AdamW(flat, lr = 1e-3, weight_decay = 1e-2)), torch.bfloat16, None
	eager = _interleaved(ca, d, make_opt, dt, lvl, False, order, shapes)
	graph = _interleaved(ca, d, make_opt, dt, lvl, True, order, shapes, max_graphs = 2)
	assert graph[3].captures == 2 and graph[3].replays >= 5 and graph[3].eager_steps >= 4, (graph[3].captures, graph[3].replays, graph[3].eager_steps)
	assert not graph[3].non_kernel_nodes and all(set(g['node_kinds']) == {'kernel'} for g in graph[3].graphs.values()), [g['node_kinds'] for g in graph[3].graphs.values()]
	bad = [(i, order[i], a, b) for i, (a, b) in enumerate(zip(eager[0], graph[0])) if a != b]
	assert not bad, bad
	assert torch.equal(eager[1], graph[1]) and eager[2] == graph[2]


def test_capture_after_a_validation_pass_still_records_every_pack_launch():
	"""fp32 compute (every packed copy is version-keyed, none is served by the optimizer's 16-bit mirror): a validation forward right before
	the step that gets captured makes the packed forward copies current, so without functional.force_repack() the graph would hold no pack
	node for them and replay with frozen weights."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	shapes = dict(A = (4, 3))
	order = list('AAAAAA')
	make_opt = lambda flat: ca.train.SGD(flat, lr = 5e-2, momentum = 0.9, weight_decay = 1e-3)
	for dt in (torch.float32, 'bf16x3'):
		eager = _interleaved(ca, d, make_opt, dt, None, False, order, shapes, validate_before_capture = {1})
		graph = _interleaved(ca, d, make_opt, dt, None, True, order, shapes, validate_before_capture = {1})
		assert graph[3].captures == 1 and graph[3].replays == 5
		assert eager[0] == graph[0], (dt, list(zip(eager[0], graph[0])))
		assert torch.equal(eager[1], graph[1])


def test_transcribe_setup_with_split_operand_inference_reproduces_the_reference_strings():
	"""convasr_amd.transcribe.setup(args.fp16 = 'bf16x3') against the vectors the reference's own transcribe body produced (tests/golden/transcribe.*):
	log-probs within 5e-4 absolute (plain bf16 inference is at ~5e-2 on this net), greedy strings identical."""
	import json
	import os
	import types
	import numpy as np
	import convasr_amd as ca
	root = os.path.dirname(os.path.abspath(__file__))
	g = np.load(os.path.join(root, 'golden', 'transcribe.npz'))
	j = json.load(open(os.path.join(root, 'golden', 'transcribe.json')))
	T_ = lambda a: torch.as_tensor(np.asarray(a))
	sd = {k[3:]: T_(g[k]) for k in g.files if k.startswith('sd/')}
	ckpt_args = dict(j['args'], alphabet = j['alphabet'], model_kwargs = dict(base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, nonlinearity = ('hardtanh', 0, 20), dilation = 2))
	args = types.SimpleNamespace(checkpoint = dict(args = dict(ckpt_args), model_state_dict = sd), device = 'cuda:0', fp16 = 'bf16x3', frontend_in_model = True, model = None, align = False)
	try:
		text_pipeline, frontend, model, generator = ca.transcribe.setup(args)
		assert model.split_dtype == torch.bfloat16 and model.backbone[1].split_inference and model.compute_dtype == torch.float32
		res = ca.transcribe.transcribe_batch(args, text_pipeline, model, generator, T_(g['wav']).unsqueeze(1), T_(g['xlen']), T_(g['begin']), T_(g['end']))
		assert torch.equal(res.olen.cpu(), T_(g['olen']))
		err = float((res.log_probs.cpu() - T_(g['log_probs'])).abs().max())
		assert err <= 5e-4, err  # measured 1.7e-4 on log-probs down to -11 (16 significant bits per operand); the exact fp32 path's bar in test_bf16_parity_gpu is 1e-4
		assert res.hyp == j['hyp'], (res.hyp, j['hyp'])
	finally:
		torch.set_grad_enabled(True)  # (transcribe.setup switches autograd off for the process, like the reference)


# ---- split forward, one-product backward ('bf16x3f' / 'f16x3f') -------------------------------------------------------------------

@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(3, 256, 384, 300, 11, 1, 5), (2, 768, 896, 200, 29, 2, 29), (3, 896, 128, 257, 1, 1, 0), (2, 192, 136, 77, 3, 1, 1)])
def test_wgrad_reads_the_hi_plane_of_a_plane_tensor_in_place(dt, shape):
	"""convasr_conv1d_wgrad_ld (frames 3 Cin elements apart) = convasr_conv1d_wgrad on the plane copied out, bit for bit (the same kernel, the same
	order of summation); outside the LDS-DMA kernel's envelope (the last shape) ops.conv1d_wgrad_hi copies the plane out itself.  The hi plane is
	the value rounded once."""
	from convasr_amd import ops
	d = torch.device('cuda:0')
	B, Cin, Cout, T, K, dil, pad = shape
	torch.manual_seed(2)
	x = ops.as_cl(torch.randn(B, Cin, T).to(d), torch.float32)
	Tout = ops.conv_out_len(T, K, 1, dil, pad)
	dy = ops.as_cl((torch.randn(B, Cout, Tout) * 1e-2).to(d), dt)
	x3 = ops.split3(x, dt, ops.SPLIT_INPUT)
	x16 = ops.split3_plane(x3)
	assert torch.equal(x16, ops.as_cl(x, dt)) and torch.equal(ops.split3_plane(x3, 2), x16)
	a, b = torch.empty(Cout, Cin, K, device = d), torch.empty(Cout, Cin, K, device = d)
	ops.conv1d_wgrad_hi(x3, dy, Cout, K, dil, pad, a)
	ops.conv1d_wgrad(x16, dy, Cout, K, 1, dil, pad, b)
	assert torch.equal(a, b)
	ref = torch.nn.grad.conv1d_weight(x16.double().cpu(), (Cout, Cin, K), dy.double().cpu(), padding = pad, dilation = dil)
	assert _rel(a, ref) <= 2e-6
	ops.conv1d_wgrad_hi(x3, dy, Cout, K, dil, pad, a, accumulate = True)  # (+)=
	assert _rel(a, 2 * ref) <= 2e-6
	# tap-major gradient memory (the training arena's) behind the same view
	c = torch.empty(K, Cout, Cin, device = d).permute(1, 2, 0)
	ops.conv1d_wgrad_hi(x3, dy, Cout, K, dil, pad, c)
	assert torch.equal(c, b)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_bn_backward_apply_rounds_dy_once_and_the_dgrad_operand_is_the_plain_16_bit_one(dt):
	from convasr_amd import ops, _lib
	d = torch.device('cuda:0')
	torch.manual_seed(7)
	B, C, T = 3, 136, 53
	y = ops.as_cl(torch.randn(B, C, T).to(d) * 3, torch.float32)
	scale, shift = (torch.rand(C) + 0.5).to(d), torch.randn(C).to(d)
	xl = torch.tensor([1.0, 0.7, 0.4], device = d)
	act = (_lib.ACT_HARDTANH, 0.0, 20.0)
	gate = torch.zeros(B * T * C // 8, dtype = torch.uint8, device = d)
	ops.bn_act(y, scale, shift, act, gate = gate, xlen = xl, dropout_p = 0.2, seed = 5, offset = 11)
	coef = torch.randn(3 * C, device = d)
	dz = ops.as_cl(torch.randn(B, C, T).to(d) * 1e-2, torch.float32)
	for g, from_dz in ((None, True), (gate, True), (None, False)):
		kw = dict(xlen = xl, dropout_p = 0.2, seed = 5, offset = 11, gate = g)
		dy = ops.bn_act_bwd_apply(dz, y, coef, from_dz, scale, shift, act, **kw)
		dy16 = ops.bn_act_bwd_apply(dz, y, coef, from_dz, scale, shift, act, planes = dt, hi_only = True, **kw)
		assert dy16.dtype == dt and dy16.shape == dy.shape and ops.is_cl(dy16) and torch.equal(dy16, ops.as_cl(dy, dt))
	# the packed operands: forward planes as ever, the dgrad operand = the ordinary 16-bit pack of the same weight
	w = (torch.randn(136, 192, 5) / 30).to(d)
	f3, d3 = ops.pack_weight_split3(w, dt)
	f1, d1 = ops.pack_weight_split3(w, dt, dgrad_planes = 1)
	assert torch.equal(f1, f3) and torch.equal(d1, ops.pack_weight(w, dt, _lib.PACK_DGRAD)) and torch.equal(d1[:, :192], d3[:, :192, :136])


@pytest.mark.parametrize('name', ['bf16x3f', 'f16x3f'])
def test_split_forward_with_one_product_backward(name):
	"""compute_dtype = 'bf16x3f' / 'f16x3f' on a small Wav2Letter: the loss is the split path's bit for bit (the forward is the same launches), every
	conv's backward is one 16-bit product per gradient (no split dgrad / wgrad launch is left, the 38-class head's included), and the gradients sit
	where the plain 16-bit path's do against the full split path's."""
	import convasr_amd as ca
	from convasr_amd import _lib
	d = torch.device('cuda:0')
	x, xlen, y, ylen = _batch(d, 6, 4)
	out = {}
	for dt in (name[:-1], name, {'bf16x3f': torch.bfloat16, 'f16x3f': torch.float16}[name]):
		model = _small(ca, d, dt)
		flat = ca.train.FlatParameters(model)
		timer = _lib.KernelTimer(only = ())
		_lib.timer = timer
		try:
			res = model(x, xlen, y = y, ylen = ylen)
			(res['loss'] * ylen[:, 0]).mean().mul(256.0).backward()  # (a fixed loss scale: fp16's output gradients underflow without one)
		finally:
			_lib.timer = None
		ca.functional.join_side_streams()
		flat.finalize_grads()
		torch.cuda.synchronize()
		out[dt] = (res['loss'].detach().clone(), {n: p._convasr_grad.clone() / 256.0 for n, p in model.named_parameters() if hasattr(p, '_convasr_grad')}, [f for f, _ in timer.sequence])
	full, mixed, half = out.values()
	assert torch.equal(full[0], mixed[0])
	assert model.compute_dtype in (torch.bfloat16, torch.float16)
	fams = mixed[2]
	assert fams.count('conv1d_igemm_v2s_kernel<x3>') == 17 + 1 + 1 and fams.count('conv1d_wgrad<x3>') == 0, (fams.count('conv1d_igemm_v2s_kernel<x3>'), fams.count('conv1d_wgrad<x3>'))  # the forwards (17 layers, the folded prologue, the head): nothing of the backward
	assert sum('conv1d_igemm_v2s_kernel' in f and '<x3>' not in f for f in fams) == 18 and fams.count('conv1d_wgrad') == 19 and 'conv1d_igemm (other variants)' not in fams, [f for f in fams if 'conv' in f]  # (a small one-tap launch is booked as memory-bound)
	assert fams.count('hbm:split3_kernel') == 2  # the folded prologue's input and the head's input
	worst = max((_rel(mixed[1][n], full[1][n]), n) for n in full[1] if float(full[1][n].abs().max()) > 0)
	worst16 = max((_rel(half[1][n], full[1][n]), n) for n in full[1] if float(full[1][n].abs().max()) > 0)
	assert worst[0] <= 3e-2 and worst[0] <= 1.5 * worst16[0], (worst, worst16)


def test_one_product_backward_in_a_dense_residual_network_and_under_graph_replay():
	"""'bf16x3f' through the residual forms (1x1 branches, fp32 dy of the grouped BN backward rounded once) -- gradients within the 16-bit bar of the
	full split path's -- and its training step replayed from a graph, bit for bit against the eager steps (dropout on)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	x, xlen, y, ylen = _batch(d, 4, 4)
	out = {}
	for dt in ('bf16x3', 'bf16x3f'):
		torch.manual_seed(4)
		fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
		model = ca.models.JasperNet(64, [38], frontend = fe, base_width = 64, kernel_sizes = [11, 13, 17], out_width_factors = [2, 3, 4], dropouts = [0.0] * 3, out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, num_subblocks = 2, dropout = 0, check_time_dim_padded = False, temporal_mask = False, compute_dtype = dt).to(d).train()
		flat = ca.train.FlatParameters(model)
		res = model(x, xlen, y = y, ylen = ylen)
		(res['loss'] * ylen[:, 0]).mean().backward()
		ca.functional.join_side_streams()
		flat.finalize_grads()
		torch.cuda.synchronize()
		out[dt] = (res['loss'].detach().clone(), {n: p._convasr_grad.clone() for n, p in model.named_parameters() if hasattr(p, '_convasr_grad')})
	a, b = out['bf16x3'], out['bf16x3f']
	assert torch.equal(a[0], b[0])
	worst = max((_rel(b[1][n], a[1][n]), n) for n in a[1] if float(a[1][n].abs().max()) > 0)
	assert worst[0] <= 3e-2, worst
	traces = []
	for graphed in (False, True):
		ca.functional.manual_seed(17)
		model = _small(ca, d, 'bf16x3f', dropout = 0.2)
		flat = ca.train.FlatParameters(model)
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed)
		batch = _batch(d, 6, 4)
		tr = [float(stepper(*batch, iteration = it)['loss']) for it in range(5)]
		torch.cuda.synchronize()
		traces.append((tr, flat.data.clone(), stepper.replays))
	assert traces[0][0] == traces[1][0] and torch.equal(traces[0][1], traces[1][1]) and traces[1][2] >= 3
