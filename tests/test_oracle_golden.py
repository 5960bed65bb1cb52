"""Pin the CPU oracle (oracle/convasr_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py).  No GPU."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import convasr_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
T = lambda a: a.detach() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a))


def close(a, b, rtol = 1e-5, atol = 1e-6):
	a, b = T(a).double(), T(b).double()
	assert a.shape == b.shape, (a.shape, b.shape)
	assert torch.allclose(a, b, rtol = rtol, atol = atol), float((a - b).abs().max())


def test_mel_filterbank_closed_form_properties(golden):
	g = golden('frontend.npz')
	W = O.mel_filterbank(16000, 512, 64)
	assert W.shape == (64, 257) and W.dtype == np.float32
	assert np.array_equal(W, g['mel_weight'][:, :, 0])
	assert (W >= 0).all()
	for row in W:  # each row is one triangle: rises then falls, contiguous support
		nz = np.nonzero(row)[0]
		assert len(nz) > 0 and np.array_equal(nz, np.arange(nz[0], nz[-1] + 1))
		peak = row.argmax()
		assert (np.diff(row[nz[0]:peak + 1]) >= 0).all() and (np.diff(row[peak:nz[-1] + 1]) <= 0).all()
	# Slaney area normalisation: integral of each triangle over Hz equals 1 (up to the 31.25 Hz bin sampling)
	area = W.sum(axis = 1) * (8000 / 256)
	assert np.allclose(area[5:], 1.0, atol = 0.12)


def test_lengths_and_mask(golden):
	g = golden('instnorm.npz')
	lengths = O.compute_output_lengths(201, T(g['xlen']))
	assert torch.equal(lengths, T(g['lengths']))
	assert torch.equal(O.temporal_mask(201, lengths), T(g['mask']).reshape(4, 201))
	assert torch.equal(O.compute_output_lengths(7, None, batch = 3), torch.full((3, ), 7))


def test_frontend(golden):
	g = golden('frontend.npz')
	kw = dict(window = T(g['window']), mel_weight = T(g['mel_weight']), mel_bias = T(g['mel_bias']), nfft = 512, hop_length = 160)
	close(O.logmel_frontend(T(g['x']), T(g['xlen']), **kw), g['feat_masked'])
	close(O.logmel_frontend(T(g['x']), None, **kw), g['feat_nomask'])
	close(O.logmel_frontend(T(g['x16']), T(g['xlen']), **kw), g['feat_int16'])
	close(O.logmel_frontend(T(g['short']), None, **kw), g['feat_short'])
	cfg = O.frontend_config()
	assert cfg == dict(win_length = 320, hop_length = 160, nfft = 512, freq_cutoff = 257)


def test_instance_norm(golden):
	g = golden('instnorm.npz')
	close(O.masked_instance_norm(T(g['x']), T(g['mask']).reshape(4, 201)), g['y_masked'])
	close(O.masked_instance_norm(T(g['x']), None), g['y_legacy'])


@pytest.mark.parametrize('ci', range(5))
def test_conv_block(golden, ci):
	case = json.load(open(os.path.join(GOLDEN, 'convblock_cases.json')))[ci]
	g = golden(f'convblock{ci}.npz')
	sd = {k[len(f'c{ci}/sd/'):]: T(g[k]).clone() for k in g.files if k.startswith(f'c{ci}/sd/')}
	sd = {'b.' + k: v for k, v in sd.items()}
	layer = dict(cin = case['cin'], cout = case['cout'], k = case['k'], stride = case['stride'], dilation = case['dilation'], repeat = case['repeat'], res = [case['cin']] * case['nres'])
	params = [k for k, v in sd.items() if v.is_floating_point() and 'running' not in k]
	for k in params:
		sd[k].requires_grad_(True)
	x = T(g['x']).clone().requires_grad_(True)
	res = [T(g[f'res{r}']) for r in range(case['nres'])]
	y = O.conv_block(x, sd, 'b', layer, T(g['frac']), res, tuple(case['nonlinearity']), case['temporal_mask'], training = True)
	close(y, g['y'], atol = 1e-5)
	y.backward(T(g['gout']))
	close(x.grad, g['gx'], rtol = 1e-4, atol = 1e-5)
	for k in params:
		close(sd[k].grad, g[f'c{ci}/grad/' + k[2:]], rtol = 1e-4, atol = 1e-4)
	for k in g.files:
		if k.startswith(f'c{ci}/sd_after/') and 'running' in k:
			close(sd['b.' + k[len(f'c{ci}/sd_after/'):]], g[k])
	with torch.no_grad():
		ye = O.conv_block(T(g['x']), sd, 'b', layer, T(g['frac']), res, tuple(case['nonlinearity']), case['temporal_mask'], training = False)
	close(ye, g['y_eval'], atol = 1e-5)


def test_ctc_torch_and_numpy_restatement(golden):
	g = golden('ctc.npz')
	lp = T(g['log_probs']).clone().requires_grad_(True)
	loss = O.ctc_loss(lp, T(g['targets']), T(g['olen']), T(g['ylen']))
	assert torch.equal(torch.isinf(loss), torch.isinf(T(g['loss'])))
	fin = ~torch.isinf(loss)
	close(loss[fin], g['loss'][fin.numpy()])
	(loss[:5] * T(g['grad_weights'])[:5]).sum().backward()
	close(lp.grad[:5], g['grad'][:5], atol = 1e-6)
	assert np.isnan(g['grad'][5]).any()  # infeasible sample: ATen leaves NaN; the training loop skips the step (train.py:769)
	# independent float64 alpha-beta recursion
	nll, grad = O.ctc_loss_numpy(g['log_probs'], g['targets'], g['olen'], g['ylen'])
	assert np.isinf(nll[5]) and nll[5] > 0
	assert np.allclose(nll[:5], g['loss'][:5], rtol = 1e-5)
	w = g['grad_weights']
	assert np.allclose(grad[:5] * w[:5, None, None], g['grad'][:5], atol = 1e-4)  # ATen accumulates in fp32; this restatement in fp64
	assert (grad[1, :, 40:] == 0).all()  # frames >= olen get exactly zero gradient


def _tiny(g, prefix = 'sd/'):
	sd = {k[len(prefix):]: T(g[k]).clone() for k in g.files if k.startswith(prefix)}
	plan = O.jasper_plan(64, [38], nonlinearity = ('hardtanh', 0, 20), dilation = 2, **O.TINY)
	return sd, plan


def test_tiny_end_to_end_and_train_steps(golden):
	g = golden('tiny_e2e.npz')
	sd, plan = _tiny(g)
	fe = dict(nfft = 512, hop_length = 160)
	bufs = {}
	args = (T(g['wav']), T(g['xlen']), T(g['y']), T(g['ylen']))
	r0 = O.train_step(sd, plan, *args, frontend = fe, momentum_buffers = bufs)
	close(r0['log_probs'], g['step0/log_probs'], rtol = 1e-4, atol = 1e-4)
	assert torch.equal(r0['olen'], T(g['step0/olen']))
	close(r0['loss_vec'], g['step0/loss_vec'], rtol = 1e-5)
	close(r0['loss'], g['step0/loss'], rtol = 1e-5)
	close(r0['entropy'], g['step0/entropy'], rtol = 1e-5)
	close(r0['grad_norm'], g['step0/grad_norm'], rtol = 1e-4)
	scale = min(1.0, 100.0 / (float(g['step0/grad_norm']) + 1e-6))  # goldens hold post-clip grads
	for k in ['decoder.0.weight', 'backbone.0.conv.0.0.weight', 'backbone.2.bn.0.weight']:
		close(r0['grads'][k], g['step0/grad/' + k], rtol = 1e-3, atol = 1e-4 * scale)
	r1 = O.train_step(sd, plan, *args, frontend = fe, momentum_buffers = bufs)
	close(r1['loss'], g['step1/loss'], rtol = 1e-4)
	close(r1['grad_norm'], g['step1/grad_norm'], rtol = 1e-3)
	for k in g.files:
		if k.startswith('sd_after2/') and 'num_batches' not in k:
			close(sd[k[len('sd_after2/'):]], g[k], rtol = 1e-3, atol = 1e-5)
	with torch.no_grad():
		ev = O.jasper_forward(sd, plan, T(g['wav']), T(g['xlen']), frontend = fe, training = False)
	close(ev['logits'], g['eval_logits'], rtol = 1e-3, atol = 1e-4)
	hyp = json.load(open(os.path.join(GOLDEN, 'tiny_e2e_hyp.json')))
	assert O.greedy_decode(ev['log_probs'], ev['olen']) == hyp['hyp']
	assert O.greedy_decode(T(g['step0/log_probs']), T(g['step0/olen'])) == hyp['hyp_step0']


def test_greedy_decode_rules(golden):
	g = golden('decode.npz')
	hyp = json.load(open(os.path.join(GOLDEN, 'decode_hyp.json')))['hyp']
	assert O.greedy_decode(T(g['log_probs']), T(g['olen'])) == hyp


def test_dense_residual_wiring(golden):
	g = golden('dense_jasper.npz')
	sd = {k[3:]: T(g[k]).clone() for k in g.files if k.startswith('sd/')}
	plan = O.jasper_plan(64, [38], base_width = 32, kernel_sizes = [11, 13], out_width_factors = [2, 3], out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, num_subblocks = 2, temporal_mask = False)
	for k in ['backbone.1.conv_residual.0.weight', 'backbone.0.conv.0.0.weight']:
		sd[k].requires_grad_(True)
	out = O.jasper_forward(sd, plan, T(g['x']), T(g['xlen']), training = True)
	close(out['logits'], g['logits'], rtol = 1e-4, atol = 1e-5)
	out['logits'].square().mean().backward()
	for k in ['backbone.1.conv_residual.0.weight', 'backbone.0.conv.0.0.weight']:
		close(sd[k].grad, g['grad/' + k], rtol = 1e-3, atol = 1e-6)


def test_wav2letter_layout_matches_reference():
	layout = json.load(open(os.path.join(GOLDEN, 'wav2letter_layout.json')))
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	sd = O.init_state_dict(plan)
	ref = {e['name']: e['shape'] for e in layout['state_dict']}
	assert {k: list(v.shape) for k, v in sd.items()} == ref
	n = sum(v.numel() for k, v in sd.items() if v.is_floating_point() and 'running' not in k)
	assert n == layout['num_params']


# ------------------------------------------------------------------------------------------------ SURVEY 8(f) "next" rows

def test_oracle_novograd_matches_reference():
	g = np.load(os.path.join(GOLDEN, 'novograd.npz'))
	for case in (0, 1):
		lr, b1, b2, eps, wd, damp, max_norm = [float(v) for v in g[f'c{case}/hyper']]
		n = len([k for k in g.files if k.startswith(f'c{case}/p0/')])
		params = [torch.from_numpy(g[f'c{case}/p0/{i}']).clone() for i in range(n)]
		state = {}
		for step in range(4):
			grads = [torch.from_numpy(g[f'c{case}/g{step}/{i}']) for i in range(n)]
			norm = O.novograd_step(params, grads, state, lr = lr, betas = (b1, b2), eps = eps, weight_decay = wd, dampening = bool(damp), max_norm = max_norm)
			assert abs(float(norm) - float(g[f'c{case}/norm{step}'])) <= 1e-5 * float(norm)
			for i in range(n):
				np.testing.assert_allclose(params[i].numpy(), g[f'c{case}/p{step + 1}/{i}'], rtol = 2e-5, atol = 2e-6)
				np.testing.assert_allclose(float(state['ema'][i]), float(g[f'c{case}/ema{step + 1}/{i}']), rtol = 2e-5)


def test_oracle_alignment_matches_reference():
	g = np.load(os.path.join(GOLDEN, 'alignment.npz'))
	for case in (0, 1, 2):
		T_ = lambda k: torch.from_numpy(g[f'c{case}/{k}'])
		al = O.ctc_alignment(T_('log_probs'), T_('targets'), T_('input_lengths'), T_('target_lengths'), blank = int(g[f'c{case}/blank']))
		assert torch.equal(al, T_('alignment')), case


def _collate_inputs(g):
	return [(torch.from_numpy(g[f'collate/in/s{k}']), torch.from_numpy(g[f'collate/in/x{k}']), torch.from_numpy(g[f'collate/in/y0_{k}']), torch.from_numpy(g[f'collate/in/y1_{k}'])) for k in range(5)]


def test_oracle_bucketing_and_collate_match_reference():
	g = np.load(os.path.join(GOLDEN, 'bucketing.npz'))
	bucket = torch.from_numpy(g['bucket'])
	for world in (1, 2):
		for epoch in (0, 1, 5):
			got = O.bucketing_schedule(bucket, 8, world, epoch)
			assert torch.equal(got, torch.from_numpy(g[f'w{world}/e{epoch}'])), (world, epoch)
	s, x, xlen, y, ylen = O.collate(_collate_inputs(g), 128, int(g['collate/speaker_missing']))
	for name, t in dict(s = s, x = x, xlen = xlen, y = y, ylen = ylen).items():
		assert torch.equal(t, torch.from_numpy(g[f'collate/{name}'])), name


def test_oracle_entropy_helpers_match_reference(golden):
	"""models.weighted_mean_entropy (models.py:660-682) and models.normalize_signal (684-686): tests/golden/make_golden_r2.py."""
	g = golden('helpers.npz')
	lp, olen = T(g['log_probs']), T(g['olen'])
	close(O.weighted_mean_entropy(lp, olen), g['wme_len'], rtol = 1e-6)
	close(O.weighted_mean_entropy(lp), g['wme_all'], rtol = 1e-6)
	close(O.weighted_mean_entropy(lp, olen, eps_id = 3), g['wme_id3'], rtol = 1e-6)
	close(O.entropy(lp, olen), g['ent_len'], rtol = 1e-6)
	close(O.normalize_signal(T(g['signal'])), g['signal_norm'], rtol = 1e-7, atol = 0)
	close(O.normalize_signal(T(g['signal']), denom_multiplier = 2.5), g['signal_norm_mult'], rtol = 1e-7, atol = 0)
