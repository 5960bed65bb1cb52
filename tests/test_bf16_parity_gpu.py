"""Full-size parity and robustness cases on the MI355X: the 16-bit storage types (bf16, fp16) against the oracle, BASELINE configs[4]
(JasperNetLarge, bucketed mixed-length batches, NovoGrad; fp16 with dynamic loss scaling as the config states, and bf16), side-stream
weight gradients, the device-side skip gate."""
import os

import numpy as np
import pytest
import torch

from oracle import convasr_oracle as O

pytestmark = pytest.mark.gpu
FE = dict(nfft = 512, hop_length = 160)
STORAGE = dict(bf16 = torch.bfloat16, f16 = torch.float16)
# per-layer bounds (relative L2 against the oracle's restatement with the same storage type, same inputs on both sides), measured numbers
# in profiles/r03_*_per_layer.json: an output that flips a rounding moves by one ulp (2^-8 bf16, 2^-11 fp16)
LAYER_BOUNDS = dict(bf16 = dict(z = 2e-4, dw = 1e-3, dbeta = 5e-4, dgamma = 5e-4, dx = 4e-3), f16 = dict(z = 8e-5, dw = 4e-4, dbeta = 1e-4, dgamma = 1e-4, dx = 1.5e-3))  # measured maxima: bf16 7.9e-5 / 4.8e-4 / 8.6e-5 / 5e-6 / 2.3e-3, fp16 2.9e-5 / 1.7e-4 / 2.7e-5 / 1.6e-6 / 7.4e-4


def _dump(name, obj):
	"""Measured numbers go to gpurun_out/ (scratch) when it exists; the ones quoted in DESIGN.md are copied to profiles/."""
	import json
	out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
	if os.path.isdir(out):
		with open(os.path.join(out, name), 'w') as f:
			json.dump(obj, f, indent = 1)


def _cos_rel(a, b):
	a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
	return float(torch.dot(a, b) / (a.norm() * b.norm())), float((a - b).norm() / b.norm())


def _wav2letter_case(ca, seed = 1, B = 4, secs = 10):
	torch.manual_seed(seed)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False)
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	x = torch.rand(B, 16000 * secs) * 2 - 1
	xlen = torch.linspace(0.5, 1, B)
	y = torch.randint(0, 37, (B, 1, 10 * secs))
	ylen = torch.tensor([[50], [60], [80], [100]])[:B]
	return model, sd, x, xlen, y, ylen


@pytest.mark.parametrize('storage', ['bf16', 'f16'])
def test_full_wav2letter_16bit_every_layer_vs_storage_oracle(storage):
	"""The dtype of the headline number (bf16) and the one BASELINE configs[4] names (fp16), at full model size, kernel by kernel: each of the 18 Conv+BN+hardtanh+mask layers of
	Wav2Letter full (4 x 10 s: 501 frames, 256..1024 channels, k = 11 / 29 dilated / 1, stride-2 prologue) and the decoder runs
	forward AND backward in bf16 on the input the fp32 oracle chain produces at that depth, against the oracle's restatement of the
	same layer with bf16 storage (operands, conv output, layer output and the gradients dy / dz rounded to bf16 where the HIP kernels
	store bf16; every sum in fp32).
	Why layer by layer: two bf16 pipelines that differ by 1e-7 anywhere do not stay 1e-7 apart.  A dense relative perturbation e
	ahead of a bf16 store turns into a fraction e / ulp of elements rounding the other way by a whole ulp, i.e. an L2 error
	sqrt(e ulp): 2e-5 -> 3e-4 -> 1e-3 -> ... -> the rounding noise itself (ulp = 2^-8).  Measured here with three layers between
	comparison points: 1-2 % in the deepest gradients.  With the same input on both sides the comparison is about the kernels:
	outputs agree to 2e-4 relative L2, gradients to 2e-3."""
	import convasr_amd as ca
	from convasr_amd import functional as Fn
	d = torch.device('cuda:0')
	model, sd, x, xlen, y, ylen = _wav2letter_case(ca)
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	torch.set_num_threads(min(os.cpu_count() or 1, 32))
	with torch.no_grad():
		feat = O.logmel_frontend(x, xlen, sd['frontend.window'], sd['frontend.mel.weight'], sd['frontend.mel.bias'], 512, 160)
		h = O.masked_instance_norm(feat, O.temporal_mask(feat.shape[-1], O.compute_output_lengths(feat.shape[-1], xlen)))
	st = STORAGE[storage]
	model.to(d).train().set_compute_dtype(st)
	bf = lambda t: t.to(st).float()
	rel = lambda a, b: float((a.detach().double().cpu() - b.detach().double()).norm() / b.detach().double().norm())
	g = torch.Generator().manual_seed(3)
	table = {}
	xlen_d = xlen.to(d)
	for i, layer in enumerate(plan['layers']):
		blk = model.backbone[i]
		for j in range(layer['repeat']):
			one = dict(layer, repeat = 1, cin = layer['cin'] if j == 0 else layer['cout'], res = [])
			src = {f'L.conv.0.0.weight': f'backbone.{i}.conv.{j}.0.weight', **{f'L.bn.0.{n}': f'backbone.{i}.bn.{j}.{n}' for n in ('weight', 'bias', 'running_mean', 'running_var')}}
			lsd = {k: sd[v].clone() for k, v in src.items()}
			for k in ('L.conv.0.0.weight', 'L.bn.0.weight', 'L.bn.0.bias'):
				lsd[k].requires_grad_(True)
			first = i == 0 and j == 0  # the stride-2 prologue: its input (the features) needs no gradient
			xq = bf(h)
			xin = xq.clone().requires_grad_(not first)
			z_ref = O.conv_block(xin, lsd, 'L', one, xlen, [], plan['nonlinearity'], plan['temporal_mask'], True, storage = st)
			# upstream gradient with a common mode, like a real one (purely zero-mean noise makes dbeta a sum of cancelling terms)
			dz = bf((0.5 * torch.randn(z_ref.shape, generator = g) + 1.0) * (torch.rand(z_ref.shape, generator = g) < 0.7))
			z_ref.backward(dz)
			conv, bn = blk.conv[j][-1], blk.bn[j]
			for p in (conv.weight, bn.weight, bn.bias):
				p.grad = None
			xg = ca.ops.as_cl(xq.to(d), st).requires_grad_(not first)
			z = Fn.ConvBnActFunction.apply(blk._cfg(j, j == layer['repeat'] - 1), xg, conv.weight, bn.weight, bn.bias, xlen_d)
			z.backward(dz.to(d))
			r = dict(z = rel(z, z_ref), dw = rel(conv.weight.grad, lsd['L.conv.0.0.weight'].grad), dgamma = rel(bn.weight.grad, lsd['L.bn.0.weight'].grad), dbeta = rel(bn.bias.grad, lsd['L.bn.0.bias'].grad))
			if not first:
				r['dx'] = rel(xg.grad, bf(xin.grad))  # the dgrad kernel stores dx in bf16
			table[f'backbone.{i}.{j}'] = {k: float(f'{v:.2e}') for k, v in r.items()}
			with torch.no_grad():  # the next layer's input comes from the fp32 chain
				h = O.conv_block(h, {k: v.detach() for k, v in lsd.items()}, 'L', one, xlen, [], plan['nonlinearity'], plan['temporal_mask'], True)
	# decoder head: bf16 operands, fp32 logits
	hq = bf(h)
	w, b = sd['decoder.0.weight'].clone().requires_grad_(True), sd['decoder.0.bias'].clone().requires_grad_(True)
	hin = hq.clone().requires_grad_(True)
	logits_ref = torch.nn.functional.conv1d(hin, O._stored_weight(w, st), b)
	dl = bf(torch.randn(logits_ref.shape, generator = g))
	logits_ref.backward(dl)
	for p in model.decoder.parameters():
		p.grad = None
	hg = ca.ops.as_cl(hq.to(d), st).requires_grad_(True)
	logits = model.decoder(hg)[0]
	logits.backward(dl.to(d))
	dec = dict(z = rel(logits, logits_ref), dx = rel(hg.grad, bf(hin.grad)), dw = rel(model.decoder[0].weight.grad, w.grad), dbeta = rel(model.decoder[0].bias.grad, b.grad))
	table['decoder'] = {k: float(f'{v:.2e}') for k, v in dec.items()}
	print(f'{storage} per-layer relative L2 vs {storage}-storage oracle (same inputs on both sides):')
	for k, v in table.items():
		print(' ', k, v)
	_dump(f'r03_{storage}_per_layer.json', table)
	# (a layer where one or two elements sit on a hardtanh boundary in one pipeline and not in the other -- backbone.4.0 here -- shows
	# it as dbeta ~1e-4 instead of ~1e-8 and a few more rounding flips in dy, hence in dw / dx)
	bound = LAYER_BOUNDS[storage]
	for k, v in table.items():
		assert all(v[name] <= bound[name] for name in v), (k, v, bound)


@pytest.mark.parametrize('storage', ['bf16', 'f16'])
def test_full_wav2letter_16bit_whole_network_deviation_is_the_storage_types_own(storage):
	"""(fp16: the same comparison with 11 significant bits; no loss scale here -- the seed gradient is 1 / B, far from fp16's limits.)
	Whole network, bf16 vs the fp32 oracle, Wav2Letter full at 4 x 10 s, dropout 0.  A random-init network of 18 batch-normed
	layers is an amplifier: in EXACT fp32 the MI355X path and the CPU oracle agree to 5e-5 in the logits but only to ~1.4e-2 in
	the first layer's weight gradient, and bf16 storage (8 significant bits) moves the logits by ~12 % and decorrelates the
	early-layer gradients -- for ANY implementation: the oracle's own bf16-storage restatement, run on the CPU, deviates from its
	fp32 self by the same amounts.  So the bar here is: (1) the CTC loss (what BASELINE's metric pins) within 2e-3 relative of the
	fp32 oracle, olen equal; (2) the MI355X bf16 path is no further from fp32 than the CPU bf16-storage restatement is (15 %
	slack), in logits and in every gradient checked; (3) the decoder gradient, which sees no amplification behind it, within 1e-2.
	Per-kernel bf16 accuracy is pinned by the teacher-forced test above; the measured numbers are in profiles/r02_bf16_parity.json."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	model, sd, x, xlen, y, ylen = _wav2letter_case(ca)
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	torch.set_num_threads(min(os.cpu_count() or 1, 32))
	kw = dict(frontend = FE, lr = 0.0, momentum = 0.0, weight_decay = 0.0, max_norm = 1e30)
	clone = lambda: {k: v.clone() for k, v in sd.items()}
	ref32 = O.train_step(clone(), plan, x, xlen, y, ylen, **kw)
	ref16 = O.train_step(clone(), plan, x, xlen, y, ylen, storage = STORAGE[storage], **kw)
	model.to(d).train().set_compute_dtype(STORAGE[storage])
	flat = ca.train.FlatParameters(model)
	out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out['loss'] * ylen.to(d)[:, 0]).mean().backward()
	flat.finalize_grads()
	assert torch.equal(out['olen'][0].cpu(), ref32['olen'])
	loss_rel = float(((out['loss'].detach().cpu() - ref32['loss_vec']).abs() / ref32['loss_vec'].abs()).max())
	assert loss_rel <= 2e-3, ('CTC loss rel', loss_rel)
	_, gpu_logits = _cos_rel(out['logits'][0], ref32['logits'])
	_, emu_logits = _cos_rel(ref16['logits'], ref32['logits'])
	assert gpu_logits <= 1.15 * emu_logits + 1e-3, (gpu_logits, emu_logits)
	params = dict(model.named_parameters())
	report = dict(loss_rel = loss_rel, logits_rel_gpu = gpu_logits, logits_rel_cpu_bf16_storage = emu_logits)
	for k in ['backbone.0.conv.0.0.weight', 'backbone.3.conv.1.0.weight', 'backbone.6.conv.0.0.weight', 'backbone.7.conv.0.0.weight', 'decoder.0.weight', 'backbone.5.bn.2.weight']:
		cos_g, rel_g = _cos_rel(params[k].grad, ref32['grads'][k])
		cos_e, rel_e = _cos_rel(ref16['grads'][k], ref32['grads'][k])
		report[k] = dict(gpu = (round(cos_g, 4), round(rel_g, 4)), cpu_bf16_storage = (round(cos_e, 4), round(rel_e, 4)))
		assert rel_g <= 1.15 * rel_e + 1e-3 and cos_g >= cos_e - 0.03, (k, report[k])
	print(f'{storage} whole-network deviation from the fp32 oracle (cosine, relative L2):', report)
	_dump(f'r03_{storage}_whole_network.json', report)
	assert report['decoder.0.weight']['gpu'][1] <= 1e-2


def _edit_distance(a, b):
	prev = list(range(len(b) + 1))
	for i, ca_ in enumerate(a, 1):
		cur = [i]
		for j, cb in enumerate(b, 1):
			cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca_ != cb)))
		prev = cur
	return prev[-1]


def test_fused_eval_greedy_strings_fp32_bf16_and_fp16_4x10s():
	"""Inference path of SURVEY 8(f1) at full size (Wav2Letter full, 4 x 10 s): batch-norm statistics re-estimated
	(reset_bn_running_stats_, models.py:726-733, then train-mode forwards: non-degenerate eval logits), fuse_conv_bn_eval, greedy
	decode.  fp32: strings IDENTICAL to the fp32 oracle's eval strings.  bf16: a random-init network decides many frames by margins
	smaller than what 8-bit storage moves the log-probs by, so the bar is (1) the argmax equals the fp32 oracle's on every frame
	the oracle decides by more than twice the observed log-prob deviation, and (2) the character error rate of the bf16 strings
	against the fp32 oracle's is no larger than that of the oracle's own bf16-storage restatement of the fused network (folded
	weights rounded to bf16, bf16 activations, fp32 sums) plus one point."""
	import convasr_amd as ca
	from convasr_amd.transcript_generators import GreedyCTCGenerator, CharTokenizerLegacy
	d = torch.device('cuda:0')
	model, sd0, x, xlen, y, ylen = _wav2letter_case(ca, seed = 2)
	xlen = torch.tensor([1.0, 0.9, 0.6, 0.75])
	B = x.shape[0]
	model.to(d)
	ca.models.reset_bn_running_stats_(model)
	with torch.no_grad():
		for _ in range(2):
			model(x.to(d), xlen.to(d))
	sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	torch.set_num_threads(min(os.cpu_count() or 1, 32))
	fused = O.fuse_conv_bn_eval(sd)
	with torch.no_grad():
		ref32 = O.jasper_forward(fused, plan, x, xlen, frontend = FE, training = False)
		ref16s = {st: O.jasper_forward(fused, plan, x, xlen, frontend = FE, training = False, storage = st) for st in STORAGE.values()}
	want32 = O.greedy_decode(ref32['log_probs'], ref32['olen'])
	want16s = {st: O.greedy_decode(r['log_probs'], r['olen']) for st, r in ref16s.items()}
	assert len(set(want32)) > 1 and all(len(w) > 20 for w in want32), want32
	cer = lambda hyp, ref: sum(_edit_distance(h, r) for h, r in zip(hyp, ref)) / sum(len(r) for r in ref)
	model.eval()
	model.fuse_conv_bn_eval()
	tok, gen = CharTokenizerLegacy(O.CHAR_LEGACY_ALPHABET), GreedyCTCGenerator()
	for dt in (torch.float32, torch.bfloat16, torch.float16):
		model.set_compute_dtype(dt)
		want16 = want16s.get(dt)
		with torch.no_grad():
			out = model(x.to(d), xlen.to(d))
		got = [t[0][0]['hyp'] if len(t[0]) else '' for t in gen.generate(tok, out['log_probs'][0], torch.zeros(B), torch.ones(B), output_lengths = out['olen'][0])]
		assert torch.equal(out['olen'][0].cpu(), ref32['olen'])
		lp32 = ref32['log_probs']
		top2 = lp32.topk(2, dim = 1).values
		dev = float((out['log_probs'][0].cpu() - lp32).abs().max())
		decisive = (top2[:, 0] - top2[:, 1]) > 2 * dev
		agree = out['log_probs'][0].argmax(dim = 1).cpu() == lp32.argmax(dim = 1)
		print(dt, 'max |log_prob - fp32 oracle|', dev, 'decisive frames', float(decisive.float().mean()), 'argmax agreement', float(agree.float().mean()), 'CER vs fp32 oracle', cer(got, want32), '(CPU restatement with the same storage type:', None if want16 is None else cer(want16, want32), ')')
		assert bool(agree[decisive].all())
		if dt == torch.float32:
			assert got == want32
		else:
			assert cer(got, want32) <= cer(want16, want32) + 0.01, (cer(got, want32), cer(want16, want32))
			assert cer(got, want16) <= cer(want16, want32) + (0.0 if dt == torch.bfloat16 else 0.01), (cer(got, want16), cer(want16, want32))  # the two 16-bit pipelines are closer to each other than either is to fp32 (fp16: both may already equal fp32's strings)


@pytest.mark.parametrize('storage', ['f16', 'bf16'])
def test_jaspernet_large_config4_bucketed_mixed_lengths_novograd(storage):
	"""BASELINE configs[4]: JasperNetLarge (models.py:1407-1409, 'Jasper 10x5': dense residuals, 277 M parameters), 32 utterances
	of 5-20 s per batch from BucketingBatchSampler + collate_gpu, NovoGrad.  f16 is the config AS STATED: the reference runs it under
	apex amp (train.py:704 -> models.py:755-762, opt_level O2), here data_parallel_and_autocast(model, optimizer, opt_level = 'O2'): fp16
	storage + MFMA, fp32 master weights, dynamic loss scaling from 2^16 -- the first iterations overflow and are skipped while the scale
	halves, exactly apex's start-up ("Gradient overflow.  Skipping step, loss scaler 0 reducing loss scale to ...").  bf16: the same
	kernels on the other 16-bit type, no scaler."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.manual_seed(1)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.JasperNetLarge(64, [38], frontend = fe, check_time_dim_padded = False).to(d).train()
	n_params = sum(p.numel() for p in model.parameters())
	assert 270e6 < n_params < 285e6, n_params
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
	if storage == 'f16':
		model, opt = ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2', keep_batchnorm_fp32 = True)
		assert model.compute_dtype == torch.float16 and flat.loss_scaler is not None and flat.loss_scaler.loss_scale() == 65536.0
		assert ca.train.amp_state_dict(opt) == dict(loss_scaler0 = dict(loss_scale = 65536.0, unskipped = 0))
	else:
		model, opt = ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2', compute_dtype = torch.bfloat16)
		assert model.compute_dtype == torch.bfloat16 and flat.loss_scaler is None
	ds = ca.datasets.SyntheticAudioTextDataset(256, min_duration = 5.0, max_duration = 20.0, seed = 11)
	sampler = ca.datasets.BucketingBatchSampler(ds, batch_size = 32, world_size = 1)
	sampler.set_epoch(0)
	plan = O.jasper_plan(64, [38], **O.JASPERNET_LARGE)
	seen = []

	def on_step(it, batch, res):
		meta, s, x, xlen, y, ylen = batch
		assert x.shape[0] == 32 and x.shape[1] % 128 == 0
		seen.append((x.shape[1], float(res['loss_cur']), float(res['grad_norm']), bool(res['skipped'])))

	w0 = flat.data.clone()
	n_it = 8 if storage == 'f16' else 2
	it = ca.train.train_epoch(model, opt, ca.datasets.gpu_batches(ds, sampler, d), sampler = sampler, iteration = 0, max_iterations = n_it, on_step = on_step)
	assert it == n_it and sampler.batch_idx == n_it
	applied = [s for s in seen if np.isfinite(s[2])]  # (an overflowed fp16 step reports a non-finite gradient norm and changes nothing)
	for T, loss, gn, skipped in seen:
		assert np.isfinite(loss) and not skipped, seen
	for T, loss, gn, skipped in applied:
		assert gn > 0, seen
	if storage == 'f16':
		amp = ca.train.amp_state_dict(opt)['loss_scaler0']
		overflowed = len(seen) - len(applied)
		print('fp16 config4: loss scale', amp, 'overflowed steps', overflowed, 'of', len(seen), [round(s[2], 3) for s in seen])
		trailing = next((i for i, s in enumerate(reversed(seen)) if not np.isfinite(s[2])), len(seen))  # clean steps since the last overflow
		assert amp['loss_scale'] == 65536.0 / 2 ** overflowed and amp['unskipped'] == trailing and len(applied) >= 2, (amp, seen)
		assert int(flat.loss_scaler.current[7]) == overflowed
		_dump('r03_fp16_config4.json', dict(amp = amp, steps = [dict(samples = s[0], loss = s[1], grad_norm = s[2] if np.isfinite(s[2]) else None) for s in seen]))
	else:
		assert len(applied) == len(seen)
	assert not torch.equal(flat.data, w0) and bool(torch.isfinite(flat.data).all())
	assert 5 * 16000 <= min(s[0] for s in seen) and max(s[0] for s in seen) <= 20 * 16000 + 128

	# output lengths of a full mixed-length batch equal the oracle's length arithmetic (models.py:611-614 after the stride-2 prologue)
	meta, s, x, xlen, y, ylen = next(iter(ca.datasets.gpu_batches(ds, sampler, d)))
	model.eval()
	with torch.no_grad():
		out = model(x, xlen)
	frames = 1 + x.shape[1] // 160
	t_out = O.conv_out_len(frames, 11, 2, 1, 5) if hasattr(O, 'conv_out_len') else (frames + 2 * 5 - 10 - 1) // 2 + 1
	assert out['logits'][0].shape[-1] == t_out
	assert torch.equal(out['olen'][0].cpu(), O.compute_output_lengths(t_out, xlen.cpu().float(), batch = x.shape[0]))


def test_jaspernet_large_dense_residual_gradients_vs_oracle_2x5s():
	"""The dense-residual wiring of JasperNetLarge at its real widths: fp32 compute, 2 x 5 s, relu, no temporal mask; logits and the
	gradients of a residual 1x1 conv fed by the FIRST block output (10 consumers), a last-sub-block conv and the prologue against
	the CPU oracle."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.manual_seed(3)
	model = ca.models.JasperNetLarge(64, [38], dropout = 0, check_time_dim_padded = False)
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	B, F = 2, 501
	g = torch.Generator().manual_seed(5)
	x = torch.randn(B, 64, F, generator = g)
	xlen = torch.tensor([1.0, 0.7])
	y = torch.randint(0, 37, (B, 1, 40), generator = g)
	ylen = torch.tensor([[40], [25]])
	plan = O.jasper_plan(64, [38], **O.JASPERNET_LARGE)
	torch.set_num_threads(min(os.cpu_count() or 1, 32))
	ref = O.train_step(sd, plan, x, xlen, y, ylen, frontend = None, lr = 0.0, momentum = 0.0, weight_decay = 0.0, max_norm = 1e30)
	model.to(d).train()
	flat = ca.train.FlatParameters(model)
	out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out['loss'] * ylen.to(d)[:, 0]).mean().backward()
	flat.finalize_grads()
	scale = float(ref['logits'].abs().max())
	err = float((out['logits'][0].detach().cpu() - ref['logits']).abs().max())
	assert err <= 1e-3 * max(scale, 1.0), ('logits', err, scale)
	loss_rel = float(((out['loss'].detach().cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
	assert loss_rel <= 1e-4, loss_rel
	params = dict(model.named_parameters())
	names = ['backbone.10.conv_residual.0.weight', 'backbone.10.conv.4.0.weight', 'backbone.5.conv_residual.2.weight', 'backbone.0.conv.0.0.weight', 'backbone.1.bn.0.weight']
	# 55 batch-normed conv layers deep, two fp32 implementations with different summation orders agree to 0.4-1.5 % in these gradients
	# (logits to 1e-3 of their range): the random-init network amplifies rounding differences layer by layer.  Against the same
	# oracle run in float64 the MI355X fp32 path is 0.26-1.9 % off and the fp32 CPU oracle 0.31-1.1 % (profiles/r02_fp64_reference.json)
	measured = {k: _cos_rel(params[k].grad, ref['grads'][k]) for k in names}
	print('JasperNetLarge 2x5s fp32 gradients vs oracle (cos, rel):', {k: (round(c, 7), float(f'{r:.3e}')) for k, (c, r) in measured.items()}, 'logits err', err, 'of', scale, 'loss rel', loss_rel)
	_dump('r04_jasper_large_grad_parity.json', dict(logits_max_abs_err = err, logits_range = scale, ctc_rel_err = loss_rel, gradients = {k: dict(cos = c, rel = r) for k, (c, r) in measured.items()}))
	# round 4, on the fp32 kernel with two-level accumulation (conv.hip TWO_LEVEL, 7.9e-8 from float64 per conv): cos 0.99993 .. 0.999995,
	# rel 3.1e-3 (the last block's convs) .. 1.2e-2 (the prologue, 55 layers of amplification below the loss); the bars leave a
	# factor ~1.6 over the measured worst case -- the fp32 CPU oracle itself sits 0.31-1.1 % from its float64 run in these gradients
	for k, (cos, rel) in measured.items():
		assert cos >= 0.9999 and rel <= 2e-2, (k, cos, rel)
	assert measured['backbone.10.conv.4.0.weight'][1] <= 6e-3 and measured['backbone.10.conv_residual.0.weight'][1] <= 6e-3, measured


def _residual_model(ca, d):
	torch.manual_seed(0)
	ca.functional.manual_seed(9)
	return ca.models.JasperNet(64, [38], base_width = 64, kernel_sizes = [11, 13], out_width_factors = [2, 2], dropouts = [0.2, 0.2], out_width_factors_large = [4, 4], residual = True, repeat = 2, num_subblocks = 2, dropout = 0.2, check_time_dim_padded = False, temporal_mask = True, compute_dtype = torch.bfloat16).to(d).train()


def test_side_stream_wgrad_is_bitwise_identical_on_a_residual_model():
	"""enable_side_stream_wgrad on a model with batch-normed residual 1x1 convs: the main conv's wgrad runs on the side stream while
	the residual wgrads of the same backward run on the main stream -- each stream has its own split-K workspace (ops.workspace is
	keyed by stream), so two steps end with bit-identical parameters to the single-stream run."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	finals = []
	for side in (False, True):
		model = _residual_model(ca, d)
		flat = ca.train.FlatParameters(model)
		model._convasr_flat = flat
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		ca.functional.enable_side_stream_wgrad(d, side)
		try:
			g = torch.Generator().manual_seed(1)
			x = torch.randn(6, 64, 400, generator = g).to(d)
			xlen = torch.tensor([1.0, 0.7, 0.45, 0.9, 0.8, 1.0], device = d)
			y = torch.randint(0, 37, (6, 1, 20), generator = g).to(d)
			ylen = torch.tensor([[20], [15], [9], [20], [12], [18]], device = d)
			for it in range(3):
				r = ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = it)
			torch.cuda.synchronize()
			assert bool(torch.isfinite(r['loss_cur']))
			finals.append(flat.data.clone())
		finally:
			ca.functional.enable_side_stream_wgrad(d, False)
	assert torch.equal(finals[0], finals[1])


def test_novograd_gated_first_iteration_leaves_no_state():
	"""A non-finite loss on the very first iteration (device-side gate): parameters, momentum and the second-moment EMAs stay
	untouched, and the next real step is the reference's FIRST step (ema = g2, not (1 - beta2) g2: optimizers.py:76-80)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')

	def make():
		torch.manual_seed(0)
		m = torch.nn.Sequential(torch.nn.Conv1d(8, 16, 3), torch.nn.Conv1d(16, 4, 1)).to(d)
		flat = ca.train.FlatParameters(m)
		return m, flat, ca.optimizers.NovoGrad(flat, lr = 1e-2, betas = (0.95, 0.98), weight_decay = 1e-3)

	def put_grads(flat, seed):
		g = torch.Generator().manual_seed(seed)
		for p in flat.params:
			p._convasr_grad.copy_(torch.randn(p.shape, generator = g).to(d))
			p._convasr_fresh = False

	m1, f1, o1 = make()
	w0 = f1.data.clone()
	put_grads(f1, 1)
	f1.clip_grad_norm_(100.0)
	o1.step(loss_gate = torch.tensor([float('nan')], device = d))
	o1.zero_grad()
	assert torch.equal(f1.data, w0) and float(o1.momentum_buffer.abs().max()) == 0
	put_grads(f1, 2)
	f1.clip_grad_norm_(100.0)
	o1.step(loss_gate = torch.tensor([1.0], device = d))
	m2, f2, o2 = make()
	put_grads(f2, 2)
	f2.clip_grad_norm_(100.0)
	o2.step(loss_gate = torch.tensor([1.0], device = d))
	assert torch.equal(f1.data, f2.data)
	ema1, ema2 = [s['_grads_ema'] for s in o1.state.values()], [s['_grads_ema'] for s in o2.state.values()]
	assert all(torch.equal(a, b) for a, b in zip(ema1, ema2))


def test_bench_launches_two_ranks_sharing_the_gpu():
	"""`python bench.py --gpus 2` end to end on this one-GPU box: the self-launched ranks share cuda:0 and exchange gradients over
	gloo (RCCL needs one GPU per rank); the real training step runs in both, rank 0's line comes back through the parent."""
	import json
	import subprocess
	import sys
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	env = dict(os.environ, CONVASR_SHARE_GPU = '1', CONVASR_DIST_BACKEND = 'gloo')
	for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
		env.pop(k, None)
	r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'], env = env, stdout = subprocess.PIPE, stderr = subprocess.PIPE, text = True, timeout = 900)
	assert r.returncode == 0, r.stderr[-3000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
	assert len(lines) == 1, lines
	out = json.loads(lines[0])
	assert out['n_gpus'] == 2 and out['dist']['world_size'] == 2 and out['dist']['backend'] == 'gloo' and out['config']['global_batch'] == 128
	assert out['value'] > 0 and np.isfinite(out['loss'])


def test_transcribe_setup_and_batch_match_the_reference_body():
	"""convasr_amd.transcribe against vectors produced by running the reference's own classes through the body of transcribe.main
	(tests/golden/make_golden_r2.py: fused-eval forward with the (log_probs, logits, olen) dict of transcribe.setup, time stamps,
	GreedyCTCGenerator with time stamps, ctc.alignment of the targets, ref segments).  fp32: log-probs to 1e-4, olen equal, every
	hyp segment's text identical and its begin / end to 1e-5 s, the alignment bit-exact, ref segments identical.  args.fp16 = 'O2':
	fp16 as under apex (and bf16 when models.AMP_DTYPE says so): character error rate of the joined strings against the fp32 reference
	<= 3 % (one or two frames flip on near-ties)."""
	import json
	import types
	import convasr_amd as ca
	root = os.path.dirname(os.path.abspath(__file__))
	g = np.load(os.path.join(root, 'golden', 'transcribe.npz'))
	j = json.load(open(os.path.join(root, 'golden', 'transcribe.json')))
	T_ = lambda a: torch.as_tensor(np.asarray(a))
	sd = {k[3:]: T_(g[k]) for k in g.files if k.startswith('sd/')}
	ckpt_args = dict(j['args'], alphabet = j['alphabet'], model_kwargs = dict(base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, nonlinearity = ('hardtanh', 0, 20), dilation = 2))
	want_segments, want_ref = j['hyp_segments'], j['ref_segments']
	for opt_level, amp_dtype in ((None, None), ('O2', torch.float16), ('O2', torch.bfloat16)):
		ca.models.AMP_DTYPE = amp_dtype or torch.float16
		args = types.SimpleNamespace(checkpoint = dict(args = dict(ckpt_args), model_state_dict = {k: v.clone() for k, v in sd.items()}), device = 'cuda:0', fp16 = opt_level, frontend_in_model = True, model = None, align = True)
		text_pipeline, frontend, model, generator = ca.transcribe.setup(args)
		assert not torch.is_grad_enabled() and not model.training and isinstance(model.backbone[0].bn[0], torch.nn.Identity)
		assert model.compute_dtype == (torch.float32 if opt_level is None else amp_dtype) and args.sample_rate == 16000
		res = ca.transcribe.transcribe_batch(args, text_pipeline, model, generator, T_(g['wav']).unsqueeze(1), T_(g['xlen']), T_(g['begin']), T_(g['end']), y = T_(g['y']), ylen = T_(g['ylen']), segment_extra_info = j['extra'])
		assert torch.equal(res.olen.cpu(), T_(g['olen']))
		if opt_level is None:
			err = float((res.log_probs.cpu() - T_(g['log_probs'])).abs().max())
			assert err <= 1e-4, err
			assert float((res.ts.cpu() - T_(g['ts'])).abs().max()) <= 1e-6
			assert torch.equal(res.alignment.cpu(), T_(g['alignment']))
			for got_list, want_list, key in ((res.hyp_segments, want_segments, 'hyp'), (res.ref_segments, want_ref, 'ref')):
				assert [len(s) for s in got_list] == [len(s) for s in want_list]
				for got, want in zip(sum(got_list, []), sum(want_list, [])):
					assert got[key] == want[key] and got['speaker'] == want['speaker'] and got['channel'] == want['channel'], (got, want)
					assert abs(got['begin'] - want['begin']) <= 1e-5 and abs(got['end'] - want['end']) <= 1e-5, (got, want)
		if opt_level is None:
			assert res.hyp == j['hyp'], (res.hyp, j['hyp'])
		else:  # bf16: a frame whose top-2 margin is inside bf16's noise may flip; character error rate against the fp32 reference strings
			cer = sum(_edit_distance(a, b) for a, b in zip(res.hyp, j['hyp'])) / sum(len(b) for b in j['hyp'])
			agree = float((res.log_probs.argmax(dim = 1).cpu() == T_(g['log_probs']).argmax(dim = 1)).float().mean())
			print(amp_dtype, 'transcribe: CER vs fp32 reference', cer, 'argmax agreement', agree)
			assert cer <= 0.03 and agree >= 0.97, (cer, agree, res.hyp, j['hyp'])
		assert all(len(h) > 20 for h in res.hyp)
	ca.models.AMP_DTYPE = torch.float16
	torch.set_grad_enabled(True)
