"""Round-3 parity cases on the MI355X: BASELINE configs[1] at its stated batch (32 x 10 s, fp32), the 'bpe' decoder's two-head loss
against vectors from the reference, the fp32 path's distance from float64 next to torch-CPU's, and one fixed batch trained by every
compute type and by the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import convasr_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
FE = dict(nfft = 512, hop_length = 160)
T_ = lambda a: torch.as_tensor(np.asarray(a))


def _dump(name, obj):
	out = os.path.join(ROOT, 'gpurun_out')
	if os.path.isdir(out):
		with open(os.path.join(out, name), 'w') as f:
			json.dump(obj, f, indent = 1)


def close(a, b, rtol, atol, what = ''):
	a, b = a.detach().double().cpu(), (b.detach().cpu() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b))).double()
	assert a.shape == b.shape, (what, a.shape, b.shape)
	err = (a - b).abs()
	tol = atol + rtol * b.abs()
	assert bool((err <= tol).all()), f'{what}: max abs err {float(err.max()):.3e}, worst excess {float((err - tol).max()):.3e}'


def test_config1_full_wav2letter_fp32_32x10s_forward_ctc_and_strings_vs_oracle():
	"""BASELINE configs[1] AS STATED: Wav2Letter full, 32 x 10 s synthetic, fp32, logmel + conv stack + CTC forward against the CPU
	oracle (BASELINE.md tolerances: logits rtol 1e-3 / atol 1e-4 of the logit range, CTC loss 1e-4 relative, output lengths equal);
	lengths 0.5 .. 1 exercise the masks.  Greedy strings: a random-init network decides some of its 32 x 503 frames by top-2 margins
	below what ANY two fp32 summation orders differ by (here ~1e-5 in the log-probs; scaling the decoder does not help, it scales
	margin and deviation alike), so: the argmax must agree on EVERY frame the oracle decides by more than twice the observed log-prob
	deviation, such frames are > 99 % of all, and the strings of at least 30 of the 32 utterances are identical (measured: 31; the
	one that differs does so in a single character that hangs on an indecisive frame).
	This is the WEAKER twin of tests/test_training_features_gpu.py::test_trained_wav2letter_32x10s_greedy_strings_identical_to_the_oracle, which runs the
	same batch shape on a TRAINED network (decisive logits) and requires 32 of 32 identical strings in eval and in batch-statistics mode."""
	import convasr_amd as ca
	from convasr_amd.transcript_generators import GreedyCTCGenerator, CharTokenizerLegacy
	torch.manual_seed(1)
	d = torch.device('cuda:0')
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False)
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	B, secs = 32, 10
	x = torch.rand(B, 16000 * secs) * 2 - 1
	xlen = torch.linspace(0.5, 1, B)
	y = torch.randint(0, 37, (B, 1, 10 * secs))
	ylen = torch.randint(40, 10 * secs + 1, (B, 1))
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	torch.set_num_threads(min(os.cpu_count() or 1, 32))
	with torch.no_grad():
		ref = O.jasper_forward(sd, plan, x, xlen, y, ylen, frontend = FE, training = True)
	model.to(d).train()
	with torch.no_grad():
		out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	assert out['logits'][0].shape == (B, 38, 503) and torch.equal(out['olen'][0].cpu(), ref['olen'])
	scale = float(ref['logits'].abs().max())
	close(out['logits'][0], ref['logits'], 1e-3, 1e-4 * max(scale, 1.0), 'logits')
	close(out['loss'], ref['loss'], 1e-4, 0, 'CTC loss')
	tok, gen = CharTokenizerLegacy(O.CHAR_LEGACY_ALPHABET), GreedyCTCGenerator()
	decode = lambda o: [t[0][0]['hyp'] if len(t[0]) else '' for t in gen.generate(tok, o['log_probs'][0], torch.zeros(B), torch.ones(B), output_lengths = o['olen'][0])]
	got, want = decode(out), O.greedy_decode(ref['log_probs'], ref['olen'])
	lp_dev = float((out['log_probs'][0].cpu() - ref['log_probs']).abs().max())
	top2 = ref['log_probs'].topk(2, dim = 1).values
	decisive = (top2[:, 0] - top2[:, 1]) > 2 * lp_dev
	agree = out['log_probs'][0].argmax(dim = 1).cpu() == ref['log_probs'].argmax(dim = 1)
	same = sum(a == b for a, b in zip(got, want))
	assert bool(agree[decisive].all()) and float(decisive.float().mean()) > 0.99 and same >= B - 2, (same, float(agree.float().mean()), float(decisive.float().mean()))
	rel = float(((out['loss'].cpu() - ref['loss']).abs() / ref['loss'].abs()).max())
	assert len(set(want)) == B and all(len(w) > 20 for w in want)
	report = dict(logits_max_abs_err = float((out['logits'][0].cpu() - ref['logits']).abs().max()), logits_range = scale, log_probs_max_abs_err = lp_dev, ctc_rel_err = rel, identical_strings = f'{same} of {B}', argmax_agreement = float(agree.float().mean()), decisive_frames = float(decisive.float().mean()))
	print('configs[1] 32x10s fp32:', report)
	_dump('r03_config1_32x10s.json', report)


@pytest.mark.parametrize('bpe_only', [False, True])
def test_bpe_decoder_two_head_loss_matches_the_reference(bpe_only):
	"""Decoder(type = 'bpe') (models.py:23-44) and the per-head CTC losses summed (models.py:316-326; bpe_only: the BPE head's alone)
	against vectors produced by the reference itself (tests/golden/make_golden_r3.py): both heads' logits and log-probs, the loss
	vector, and gradients in both heads and in the shared encoder."""
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'bpe_decoder.npz'))
	d = torch.device('cuda:0')
	model = ca.models.JasperNet(64, [38, 48], decoder_type = 'bpe', base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, check_time_dim_padded = False, nonlinearity = ('hardtanh', 0, 20), dilation = 2, dropout = 0, bpe_only = bpe_only)
	model.load_state_dict({k[3:]: T_(g[k]) for k in g.files if k.startswith('sd/')})
	model.to(d).train()
	x, xlen, y, ylen = (T_(g[k]).to(d) for k in ('x', 'xlen', 'y', 'ylen'))
	out = model(x, xlen, y = y, ylen = ylen)
	tag = 'bpe_only' if bpe_only else 'both'
	for i in range(2):
		close(out['logits'][i], g[f'logits{i}'], 1e-3, 1e-4, f'logits head {i}')
		close(out['log_probs'][i], g[f'log_probs{i}'], 1e-3, 1e-4, f'log_probs head {i}')
		assert torch.equal(out['olen'][i].cpu(), T_(g[f'olen{i}']))
	close(out['loss'], g[f'{tag}/loss'], 1e-4, 1e-5, 'loss')
	(out['loss'] * ylen[:, 0]).mean().backward()
	params = dict(model.named_parameters())
	for k in [n[len(tag) + 6:] for n in g.files if n.startswith(tag + '/grad/')]:
		ref = g[f'{tag}/grad/{k}']
		got = params[k].grad if params[k].grad is not None else torch.zeros_like(params[k])
		close(got, ref, 5e-3, 5e-3 * float(np.abs(ref).max()) + 1e-7, 'grad ' + k)


def test_fp32_conv_accumulation_is_no_further_from_float64_than_torch_cpu():
	"""Why the round-2 fp32 path sat 1.7-3x further from float64 than torch-CPU fp32 (profiles/r02_fp64_reference.json) and what
	was done: a conv output is a sum of Cin K products; one fp32 MFMA accumulator chain over all of them (8448 terms for 768
	channels, K = 11) random-walks to ~sqrt(n) / 2 ulp, where the CPU's blocked / vectorised sum keeps dozens of short chains (one
	chain per 32-channel slab still measured 1.16e-6 against the CPU's 2.1e-7).  The fp32 kernel now closes its chain after every
	(slab, tap) step -- 32 products -- and keeps the running total in fp64 (conv.hip, TWO_LEVEL).  Here: the largest 11-tap forward
	layer of Wav2Letter, fp32, against the same conv in float64; the MI355X error must not exceed torch-CPU fp32's."""
	from convasr_amd import ops, _lib
	torch.manual_seed(4)
	d = torch.device('cuda:0')
	B, Cin, Cout, T, K = 2, 768, 768, 300, 11
	x = torch.randn(B, Cin, T).clamp_(0, 20)  # (non-negative, like the hardtanh(0, 20) activations the layer sees)
	w = torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
	torch.set_num_threads(min(os.cpu_count() or 1, 16))
	ref64 = torch.nn.functional.conv1d(x.double(), w.double(), padding = K // 2)
	cpu32 = torch.nn.functional.conv1d(x, w, padding = K // 2)
	y = ops.conv1d(ops.as_cl(x.to(d)), ops.pack_weight(w.to(d), torch.float32, _lib.PACK_FWD), Cout, K, 1, 1, K // 2)
	rel = lambda a: float((a.double().cpu() - ref64).norm() / ref64.norm())
	e_gpu, e_cpu = rel(y), rel(cpu32)
	print('fp32 conv 768->768 k=11 vs float64: MI355X', e_gpu, 'torch-CPU', e_cpu)
	_dump('r03_fp32_conv_vs_fp64.json', dict(mi355x_fp32 = e_gpu, cpu_fp32 = e_cpu, shape = [B, Cin, Cout, T, K]))
	assert e_gpu <= 1.1 * e_cpu and e_gpu <= 3e-7, (e_gpu, e_cpu)


def _first_below(traj, level):
	return next((i for i, v in enumerate(traj) if v <= level), None)


def test_every_compute_type_converges_like_the_fp32_oracle_on_one_fixed_batch():
	"""Does 16-bit storage TRAIN?  Wav2Letter full, one fixed batch of 8 x 8 s (lengths 0.6 .. 1, dropout 0), SGD lr 1e-3 / momentum
	0.9 / weight decay 1e-3 / clip 100 -- a stable setting: the fp32 CPU oracle falls from 15.9 through the blank-collapse plateau
	(~3.5, steps 6-15) to < 0.01 by step 33 -- trained for 45 applied steps by the oracle (CPU, fp32) and by the MI355X path in
	fp32, bf16, fp16 (fp16 under apex's dynamic loss scaling from 2^16: overflowed steps are skipped and not counted) and bf16x3f (split-operand
	forward, one 16-bit product per gradient in the backward).
	Past the plateau the problem is memorised at ~2x per step, so two runs that are one step apart differ by 2x in loss: 'within 5 %
	at the end' is meaningless there (the oracle's own bf16-storage restatement on the CPU trails its fp32 self by 1-4 steps and is
	1.8x above it at step 44, scratch history in profiles/r03_convergence.json).  The bar is therefore in STEPS: every compute type
	(1) tracks the oracle within 3 % (bf16: 8 %) over steps 0-6 (the descent into the plateau, before trajectories decorrelate), (2) reaches
	loss <= 1.0 / <= 0.1 / <= 0.01 no more than 3 / 5 / 6 steps after the oracle does, and (3) ends below 0.01."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	g = torch.Generator().manual_seed(7)
	B, secs, steps = 8, 8, 45
	x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
	xlen = torch.linspace(0.6, 1, B)
	y = torch.randint(0, 37, (B, 1, 10 * secs), generator = g)
	ylen = torch.randint(40, 10 * secs + 1, (B, 1), generator = g)
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	sd0 = O.init_state_dict(plan, seed = 1, frontend = O.frontend_config())
	traj = {}
	torch.set_num_threads(min(os.cpu_count() or 1, 16))
	sd, bufs = {k: v.clone() for k, v in sd0.items()}, {}
	traj['oracle_fp32'] = [float(O.train_step(sd, plan, x, xlen, y, ylen, frontend = FE, lr = 1e-3, momentum_buffers = bufs)['loss_cur']) for _ in range(steps)]
	xd, xlen_d, yd, ylen_d = x.to(d), xlen.to(d), y.to(d), ylen.to(d)
	skipped = {}
	for name, dt in (('mi355x_fp32', torch.float32), ('mi355x_bf16', torch.bfloat16), ('mi355x_fp16', torch.float16), ('mi355x_bf16x3f', 'bf16x3f')):  # (bf16x3f: the split-operand forward with a one-product backward, held to the fp32 path's bars)
		fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
		model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.0, check_time_dim_padded = False)
		assert not model.load_state_dict(sd0, strict = False).missing_keys
		model.to(d).train()
		flat = ca.train.FlatParameters(model)
		opt = ca.train.SGD(flat, lr = 1e-3, momentum = 0.9, weight_decay = 1e-3)
		if dt == torch.float16:
			ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2')
		else:
			model.set_compute_dtype(dt)
		losses, it = [], 0
		while len(losses) < steps and it < steps + 12:
			r = ca.train.train_step(model, opt, xd, xlen_d, yd, ylen_d, iteration = it)
			it += 1
			if np.isfinite(float(r['grad_norm'])):  # (fp16: an overflowed step changed nothing; the same loss comes again)
				losses.append(float(r['loss_cur']))
		traj[name], skipped[name] = losses, it - len(losses)
		assert len(losses) == steps, (name, it, len(losses))
		del model, flat, opt
	ref = traj['oracle_fp32']
	marks = {level: _first_below(ref, level) for level in (1.0, 0.1, 0.01)}
	report = dict(setting = f'Wav2Letter full, {B}x{secs}s fixed batch, dropout 0, SGD lr 1e-3 momentum 0.9 wd 1e-3 clip 100, {steps} applied steps', trajectories = {k: [float(f'{v:.5g}') for v in t] for k, t in traj.items()}, overflowed_steps_skipped = skipped, first_step_at_or_below = {k: {str(level): _first_below(t, level) for level in (1.0, 0.1, 0.01)} for k, t in traj.items()})
	print(json.dumps(report['first_step_at_or_below']), 'skipped', skipped, 'final', {k: t[-1] for k, t in traj.items()})
	_dump('r03_convergence.json', report)
	assert all(m is not None for m in marks.values()), marks
	for name, t in traj.items():
		if name == 'oracle_fp32':
			continue
		for i in range(7):  # (measured: fp32 and fp16 within 0.4 %, bf16 within 6.1 % on the steepest step)
			assert abs(t[i] - ref[i]) <= (0.08 if name == 'mi355x_bf16' else 0.03) * ref[i], (name, i, t[i], ref[i])
		for level, slack in ((1.0, 3), (0.1, 5), (0.01, 6)):
			hit = _first_below(t, level)
			assert hit is not None and hit <= marks[level] + slack, (name, level, hit, marks[level])
		assert t[-1] <= 0.01, (name, t[-1])


def test_dropout_masks_of_different_seeds_are_not_permutations_of_each_other():
	"""The counter-based dropout generator under seeds that differ in ONE bit (ranks of a data-parallel job seed it 1 + rank): the
	keep patterns of the 8-element blocks of one seed must not reappear, block-index XOR-permuted, under the other (round 2's
	hash(counter ^ key) did exactly that for seeds equal in their low two bits).  For every bit 0..9 and every XOR shift d < 64 the
	fraction of equal block patterns stays near chance (p = 0.2: sum of squared pattern probabilities = 0.68^8 = 4.6 %)."""
	from convasr_amd import ops
	d = torch.device('cuda:0')
	B, C, T = 1, 64, 4096
	y = ops.as_cl(torch.ones(B, C, T, device = d))
	act = ops.act_args(('relu', ))

	def patterns(seed):
		z = ops.bn_act(y, None, None, act, dropout_p = 0.2, seed = seed, offset = 0)
		keep = (z.permute(0, 2, 1).reshape(-1, 8) != 0).to(torch.int32)  # element index = (b T + t) C + c: memory order
		return (keep * (1 << torch.arange(8, device = d, dtype = torch.int32))).sum(dim = 1)

	base = patterns(1)
	n = base.numel()
	idx = torch.arange(n, device = d)
	assert abs(float((base != 0).float().mean()) - (1 - 0.2 ** 8)) < 1e-3
	worst = 0.0
	for bit in range(10):
		other = patterns(1 ^ (1 << bit))
		for shift in range(64):
			worst = max(worst, float((base == other[idx ^ shift]).float().mean()))
	print('dropout cross-seed: worst block-pattern agreement over 10 seed bits x 64 XOR shifts', worst)
	assert worst < 0.08, worst


def test_prepacked_dgrad_weights_leave_training_bitwise_unchanged():
	"""functional.prepack_dgrad_weights (the backward pass's transposed weight copies, run on a side stream under the CTC recursion)
	against packing them in line: three training steps of a bf16 model end with bit-identical parameters."""
	import convasr_amd as ca
	from convasr_amd import functional as Fn
	d = torch.device('cuda:0')
	g = torch.Generator().manual_seed(3)
	x = (torch.rand(6, 16000 * 3, generator = g) * 2 - 1).to(d)
	xlen = torch.tensor([1.0, 0.7, 0.45, 0.9, 0.8, 1.0], device = d)
	y = torch.randint(0, 37, (6, 1, 20), generator = g).to(d)
	ylen = torch.tensor([[20], [15], [9], [20], [12], [18]], device = d)
	finals = []
	for prepack in (True, False):
		prev, Fn.PREPACK = Fn.PREPACK, prepack
		try:
			torch.manual_seed(0)
			ca.functional.manual_seed(9)
			fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
			model = ca.models.JasperNet(64, [38], base_width = 128, kernel_sizes = [11, 13], out_width_factors = [2, 2], dropouts = [0.2, 0.2], out_width_factors_large = [4, 4], residual = True, repeat = 2, num_subblocks = 1, dropout = 0.2, frontend = fe, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d).train()
			flat = ca.train.FlatParameters(model)
			opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
			for it in range(3):
				r = ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = it)
			torch.cuda.synchronize()
			assert bool(torch.isfinite(r['loss_cur']))
			finals.append(flat.data.clone())
		finally:
			Fn.PREPACK = prev
	assert torch.equal(finals[0], finals[1])


def test_stft_conv_mode_frontend_matches_the_reference_and_loads_its_checkpoint():
	"""LogFilterBankFrontend(stft_mode = 'conv') (models.py:548-561): the reference's state dict -- with the windowed DFT basis as
	`stft.weight` -- loads with no missing or unexpected key, the basis built here equals the reference's, and the features of the
	fused FFT kernel equal the reference's conv-formulation features (tests/golden/make_golden_r3.py) within the frontend tolerance."""
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'frontend_conv.npz'))
	d = torch.device('cuda:0')
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window', stft_mode = 'conv')
	ref_sd = {k[3:]: T_(g[k]) for k in g.files if k.startswith('sd/')}
	assert set(ref_sd) == set(fe.state_dict()) and 'stft.weight' in ref_sd
	close(fe.stft.weight, ref_sd['stft.weight'], 1e-6, 1e-6, 'windowed DFT basis')
	res = fe.load_state_dict(ref_sd)
	assert not res.missing_keys and not res.unexpected_keys
	fe.to(d)
	sig, lens = T_(g['signal']).to(d), T_(g['lens']).to(d)
	feat = fe(sig, mask = ca.models.temporal_mask(sig, lens))
	close(feat, g['feat'], 5e-4, 5e-4, 'features: FFT kernel vs the conv formulation')


def test_inplace_configs_compute_the_plain_values():
	"""inplace = True (InplaceBatchNorm1d + the invertible activation, models.py:357-433; the configs JasperNetBigInplace and
	Wav2LetterDenseNoDilationInplace): a memory trick of the reference -- same statistics, same affine map, act(y + residuals) then
	dropout.  A dense-residual leaky-relu net built with inplace = True against the oracle's plain restatement: logits, loss, gradients;
	the state-dict keys equal those of the same net without the flag (InplaceBatchNorm1d subclasses nn.BatchNorm1d)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.manual_seed(5)
	kw = dict(base_width = 32, kernel_sizes = [11, 13], out_width_factors = [2, 2], dropouts = [0.2, 0.2], out_width_factors_large = [2, 2], residual = 'dense', repeat = 2, num_subblocks = 1, check_time_dim_padded = False, nonlinearity = ('leaky_relu', 0.01), dropout = 0, temporal_mask = False)
	model = ca.models.JasperNet(64, [38], inplace = True, **kw)
	assert list(model.state_dict()) == list(ca.models.JasperNet(64, [38], **kw).state_dict()) and model.backbone[1].activation.invertible
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	g = torch.Generator().manual_seed(3)
	x = torch.randn(3, 64, 121, generator = g)
	xlen = torch.tensor([1.0, 0.7, 0.85])
	y = torch.randint(0, 37, (3, 1, 10), generator = g)
	ylen = torch.tensor([[10], [6], [8]])
	plan = O.jasper_plan(64, [38], **{k: v for k, v in kw.items() if k not in ('check_time_dim_padded', 'dropout', 'dropouts')})
	ref = O.train_step(sd, plan, x, xlen, y, ylen, frontend = None, lr = 0.0, momentum = 0.0, weight_decay = 0.0, max_norm = 1e30)
	model.to(d).train()
	flat = ca.train.FlatParameters(model)
	out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out['loss'] * ylen.to(d)[:, 0]).mean().backward()
	flat.finalize_grads()
	close(out['logits'][0], ref['logits'], 1e-3, 1e-4, 'logits')
	close(out['loss'], ref['loss_vec'], 1e-4, 1e-5, 'loss')
	params = dict(model.named_parameters())
	for k in ['backbone.0.conv.0.0.weight', 'backbone.1.conv_residual.0.weight', 'backbone.2.bn.1.weight', 'decoder.0.bias']:
		r = ref['grads'][k]
		close(params[k].grad, r, 5e-3, 5e-3 * float(r.abs().max()) + 1e-7, 'grad ' + k)
	assert ca.models.JasperNetBigInplace(64, [38]).backbone[3].activation.nonlinearity == ('leaky_relu', 0.01)
	with pytest.raises(ca._lib.ConvasrHipError):
		ca.models.distributed_data_parallel_and_autocast(model, 0, synchronize_bn = True)  # training mode: cross-GPU statistics are not implemented (the reference's training path does not ask for them)


@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
@pytest.mark.parametrize('case', [(3, 64, 64, 141, 11, 16, 1), (2, 256, 384, 97, 13, 128, 1), (2, 128, 128, 50, 25, 32, 1), (2, 64, 128, 61, 5, 16, 2)])
def test_grouped_conv1d_kernels_vs_torch(case, dtype):
	"""csrc/grouped.hip (the grouped half of the separable block, models.py:50-64): forward with bias + ReLU, input gradient through the
	ReLU, weight / bias gradients against torch's grouped F.conv1d on the CPU, reference-layout and tap-major (training arena) weights."""
	from convasr_amd import ops
	B, Cin, Cout, T, K, G, stride = case
	d = torch.device('cuda:0')
	torch.manual_seed(sum(case))
	dt = dict(f32 = torch.float32, bf16 = torch.bfloat16, f16 = torch.float16)[dtype]
	x = torch.randn(B, Cin, T).to(dt).float().requires_grad_(True)
	w = (torch.randn(Cout, Cin // G, K) / (Cin // G * K) ** 0.5).requires_grad_(True)
	bias = torch.randn(Cout).requires_grad_(True)
	pad = K // 2
	y = torch.nn.functional.conv1d(x, w, bias, stride = stride, padding = pad, groups = G).relu()
	dy = torch.randn_like(y).to(dt).float()
	y.backward(dy)
	rt, at = dict(f32 = (1e-4, 2e-5), bf16 = (4e-3, 5e-5), f16 = (6e-4, 5e-5))[dtype]
	for kmajor in (False, True):
		wd = w.detach().to(d)
		if kmajor:  # (Cout, cgi, K) view of [K][Cout][cgi] memory, like FlatParameters keeps conv weights
			wd = wd.permute(2, 0, 1).contiguous().permute(1, 2, 0)
		xd, dyd = ops.as_cl(x.detach().to(d), dt), ops.as_cl(dy.to(d), dt)
		yd = ops.grouped_conv1d(xd, wd, bias.detach().to(d), G, stride, pad, relu = True)
		close(yd.float(), y, rt, at, 'forward')
		dw = torch.full_like(wd, 7.0)
		db = torch.full((Cout, ), 7.0, device = d)
		ops.grouped_conv1d_wgrad(xd, dyd, yd, dw, db, G, stride, pad)
		# (the ReLU gate comes from the stored 16-bit output: identical to the fp32 reference's except where |pre-activation| < one rounding)
		close(dw, w.grad, 2e-3, 2e-3 * float(w.grad.abs().max()), 'wgrad')
		close(db, bias.grad, 2e-3, 2e-3 * float(bias.grad.abs().max()), 'dbias')
		ops.grouped_conv1d_wgrad(xd, dyd, yd, dw, db, G, stride, pad, accumulate = True)
		close(dw, 2 * w.grad, 2e-3, 4e-3 * float(w.grad.abs().max()), 'wgrad accumulate')
		if stride == 1:
			dx = ops.grouped_conv1d_dgrad(dyd, yd, wd, Cin, T, G, stride, pad)
			close(dx.float(), x.grad, max(rt, 2e-3), 2e-3 * float(x.grad.abs().max()), 'dgrad')


def test_separable_jaspernet_matches_the_reference():
	"""JasperNet(separable = True) (models.py:50-64; JasperNetSeparable is this with 128 groups) against vectors from the reference
	(tests/golden/make_golden_r3.py): the reference's state dict (grouped weight + bias at index 0, 1x1 weight at index 2) loads
	unchanged; train-mode logits / loss / gradients, running statistics, and eval logits after fuse_conv_bn_eval."""
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'separable.npz'))
	d = torch.device('cuda:0')
	kw = dict(base_width = 32, kernel_sizes = [11, 13], out_width_factors = [2, 3], dropouts = [0.2, 0.2], out_width_factors_large = [4, 4], residual = True, repeat = 2, num_subblocks = 1, check_time_dim_padded = False, dropout = 0, separable = True, groups = 16)
	model = ca.models.JasperNet(64, [38], **kw)
	sd = {k[3:]: T_(g[k]) for k in g.files if k.startswith('sd/')}
	assert list(sd) == list(model.state_dict())
	model.load_state_dict(sd)
	model.to(d).train()
	flat = ca.train.FlatParameters(model)
	x, xlen, y, ylen = (T_(g[k]).to(d) for k in ('x', 'xlen', 'y', 'ylen'))
	out = model(x, xlen, y = y, ylen = ylen)
	close(out['logits'][0], g['logits'], 1e-3, 1e-4, 'logits')
	close(out['loss'], g['loss'], 1e-4, 1e-5, 'loss')
	(out['loss'] * ylen[:, 0]).mean().backward()
	flat.finalize_grads()
	params = dict(model.named_parameters())
	for k in [n[5:] for n in g.files if n.startswith('grad/')]:
		ref = g['grad/' + k]
		close(params[k].grad, ref, 5e-3, 5e-3 * float(np.abs(ref).max()) + 1e-7, 'grad ' + k)
	for k in [n[9:] for n in g.files if n.startswith('sd_after/')]:
		close(model.state_dict()[k], g['sd_after/' + k], 1e-3, 1e-5, k)
	model.eval()
	model.fuse_conv_bn_eval()
	with torch.no_grad():
		ev = model(x, xlen)
	close(ev['logits'][0], g['eval_logits'], 1e-3, 1e-3, 'eval logits after fuse_conv_bn_eval')
	assert sum(p.numel() for p in ca.models.JasperNetSeparable(64, [38]).parameters()) > 0
