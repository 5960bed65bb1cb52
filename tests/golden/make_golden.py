"""Generate golden vectors by running the REFERENCE itself (authoring container only).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

/root/reference is imported read-only with empty stub modules for the three imports that are not installed here
(onnxruntime, apex, librosa; soundfile for the transcript helpers).  The only function the stubs must really
provide is librosa.filters.mel (models.py:522): it is supplied by oracle.convasr_oracle.mel_filterbank and the
resulting 64x257 matrix is committed in frontend.npz so every consumer shares the same constants.

Nothing from the reference is copied: the fixtures are inputs and outputs only.
"""
import os
import sys
import types
import json
import importlib.machinery

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = '/root/reference'
sys.path.insert(0, ROOT)

from oracle import convasr_oracle as O


def import_reference():
	for name in ['onnxruntime', 'apex', 'librosa', 'librosa.filters', 'librosa.util', 'soundfile']:
		if name not in sys.modules:
			mod = types.ModuleType(name)
			mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
			sys.modules[name] = mod
	librosa = sys.modules['librosa']
	librosa.filters = sys.modules['librosa.filters']
	librosa.util = sys.modules['librosa.util']
	librosa.filters.mel = lambda sr, n_fft, n_mels = 128, fmin = 0.0, fmax = None: O.mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
	sys.path.insert(0, REFERENCE)
	import models
	import transcript_generators
	import text_tokenizers
	return models, transcript_generators, text_tokenizers


def npz(name, **arrays):
	out = {}
	for k, v in arrays.items():
		if torch.is_tensor(v):
			v = v.detach().cpu().numpy()
		out[k] = np.asarray(v)
	path = os.path.join(HERE, name)
	np.savez_compressed(path, **out)
	print(name, len(out), 'arrays', os.path.getsize(path) // 1024, 'KB')


def sd_numpy(model, prefix = 'sd/', only = None):
	return {prefix + k: v.detach().cpu().clone().numpy() for k, v in model.state_dict().items() if only is None or any(o in k for o in only)}


def main():
	models, transcript_generators, text_tokenizers = import_reference()
	torch.manual_seed(1)
	torch.set_num_threads(8)

	# ---------------------------------------------------------------- frontend (models.py:486-603)
	B, T = 4, 6400
	fe = models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	x = torch.rand(B, T) * 2 - 1
	x[1] *= 0.3
	xlen = torch.tensor([1.0, 0.9, 0.5, 0.75])
	mask = models.temporal_mask(x, models.compute_output_lengths(x, xlen))
	with torch.no_grad():
		feat_masked = fe(x, mask = mask)
		feat_nomask = fe(x)
		x16 = (x * 20000).to(torch.int16)
		feat_int16 = fe(x16, mask = mask)
		short = torch.rand(2, 200) * 2 - 1  # T <= pad: constant left pad branch (models.py:578)
		feat_short = fe(short)
	npz('frontend.npz', x = x, xlen = xlen, x16 = x16, short = short, window = fe.window, mel_weight = fe.mel.weight, mel_bias = fe.mel.bias, feat_masked = feat_masked, feat_nomask = feat_nomask, feat_int16 = feat_int16, feat_short = feat_short)

	# ---------------------------------------------------------------- lengths / instance norm (models.py:611-619, 688-719)
	feats = torch.randn(4, 64, 201) * 2 - 5
	norm = models.MaskedInstanceNorm1d(64, affine = False, eps = torch.finfo(torch.float16).tiny, track_running_stats = False, temporal_mask = True, legacy = True)
	fmask = models.temporal_mask(feats, models.compute_output_lengths(feats, xlen))
	npz('instnorm.npz', x = feats, xlen = xlen, lengths = models.compute_output_lengths(feats, xlen), mask = fmask, y_masked = norm(feats, mask = fmask), y_legacy = norm(feats, mask = None))

	# ---------------------------------------------------------------- ConvBn1d blocks (models.py:80-151)
	cases = []
	for ci, (cin, cout, k, stride, dil, rep, nonlin, nres, tmask) in enumerate([
		(64, 96, 11, 2, 1, 1, ('hardtanh', 0, 20), 0, True),
		(96, 96, 11, 1, 1, 3, ('hardtanh', 0, 20), 0, True),
		(64, 128, 29, 1, 2, 1, ('hardtanh', 0, 20), 0, True),
		(64, 64, 13, 1, 1, 2, ('relu', ), 2, False),
		(64, 32, 1, 1, 1, 1, ('leaky_relu', 0.01), 0, True),
	]):
		torch.manual_seed(10 + ci)
		blk = models.ConvBn1d(num_channels = (cin, cout), kernel_size = k, stride = stride, dilation = dil, repeat = rep, nonlinearity = nonlin, temporal_mask = tmask, num_channels_residual = [cin] * nres)
		for bn in list(blk.bn) + [b for b in blk.bn_residual]:
			bn.weight.data.uniform_(0.5, 1.5)
			bn.bias.data.uniform_(-0.5, 0.5)
		blk.train()
		xin = torch.randn(3, cin, 77) * 1.5
		frac = torch.tensor([1.0, 0.6, 0.83])
		xin_g = xin.clone().requires_grad_(True)
		res = [torch.randn(3, cin, 77) for _ in range(nres)]
		sd0 = sd_numpy(blk, f'c{ci}/sd/')
		yout = blk(xin_g, lengths_fraction = frac, residual = res)
		gout = torch.randn_like(yout)
		yout.backward(gout)
		grads = {f'c{ci}/grad/{n}': p.grad for n, p in blk.named_parameters()}
		sd1 = sd_numpy(blk, f'c{ci}/sd_after/', only = ['running_', 'num_batches'])
		blk.eval()
		with torch.no_grad():
			yeval = blk(xin, lengths_fraction = frac, residual = res)
		cases.append(dict(cin = cin, cout = cout, k = k, stride = stride, dilation = dil, repeat = rep, nonlinearity = list(nonlin), nres = nres, temporal_mask = tmask))
		npz(f'convblock{ci}.npz', x = xin, frac = frac, y = yout, gout = gout, gx = xin_g.grad, y_eval = yeval, **{f'res{r}': t for r, t in enumerate(res)}, **sd0, **sd1, **grads)
	json.dump(cases, open(os.path.join(HERE, 'convblock_cases.json'), 'w'), indent = 1)

	# ---------------------------------------------------------------- CTC (models.py:323 -> F.ctc_loss)
	torch.manual_seed(3)
	Bc, C, Tc, S = 6, 38, 60, 12
	lp = torch.randn(Bc, C, Tc).log_softmax(dim = 1).requires_grad_(True)
	y = torch.randint(0, 37, (Bc, S))
	y[0, 3] = y[0, 2]  # repeated label
	y[1, :6] = y[1, 0]  # long run of repeats
	ylen = torch.tensor([12, 12, 5, 1, 0, 12])
	olen = torch.tensor([60, 40, 60, 33, 20, 14])  # last: infeasible (14 < 12 + repeats) unless no repeats... forced inf below
	y[5, 1::2] = y[5, 0::2]  # 6 repeats -> needs 18 frames > 14 -> +inf
	loss = torch.nn.functional.ctc_loss(lp.permute(2, 0, 1), y, olen, ylen, blank = C - 1, reduction = 'none')
	gw = torch.tensor([1.0, 0.5, 2.0, 1.0, 1.0, 0.0])
	(loss[:5] * gw[:5]).sum().backward()
	npz('ctc.npz', log_probs = lp, targets = y, olen = olen, ylen = ylen, loss = loss, grad_weights = gw, grad = lp.grad)

	# ---------------------------------------------------------------- tiny model end to end (BASELINE config 1) + train step
	torch.manual_seed(1)
	fe = models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	tiny = models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, frontend = fe, check_time_dim_padded = False, nonlinearity = ('hardtanh', 0, 20), dilation = 2)
	tiny.train()
	Bt, Tt = 4, 32000
	wav = torch.rand(Bt, Tt) * 2 - 1
	wlen = torch.tensor([1.0, 0.9, 0.5, 0.75])
	yy = torch.randint(0, 37, (Bt, 1, 20))
	yylen = torch.tensor([[20], [18], [9], [14]])
	sd0 = sd_numpy(tiny)
	opt = torch.optim.SGD(tiny.parameters(), lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	steps = {}
	gen = transcript_generators.GreedyCTCGenerator()
	tok = text_tokenizers.CharTokenizerLegacy(O.CHAR_LEGACY_ALPHABET)
	first = lambda tt: [t[0][0]['hyp'] if len(t[0]) else '' for t in tt]
	for it in range(2):
		out = tiny(wav, wlen, y = yy, ylen = yylen)
		loss = (out['loss'] * yylen[:, 0]).mean()
		ent = models.entropy(out['log_probs'][0], out['olen'][0], dim = 1).mean()
		opt.zero_grad()
		loss.backward()
		gn = torch.nn.utils.clip_grad_norm_(tiny.parameters(), 100)
		if it == 0:
			hyp_step0 = first(gen.generate(tok, out['log_probs'][0].detach(), begin = torch.zeros(Bt), end = torch.ones(Bt), output_lengths = out['olen'][0]))
			steps.update({'step0/logits': out['logits'][0], 'step0/log_probs': out['log_probs'][0], 'step0/olen': out['olen'][0], 'step0/loss_vec': out['loss'], 'step0/grad/decoder.0.weight': tiny.decoder[0].weight.grad, 'step0/grad/backbone.0.conv.0.0.weight': tiny.backbone[0].conv[0][0].weight.grad, 'step0/grad/backbone.2.bn.0.weight': tiny.backbone[2].bn[0].weight.grad})
		steps.update({f'step{it}/loss': loss, f'step{it}/entropy': ent, f'step{it}/grad_norm': gn})
		opt.step()
	sd2 = sd_numpy(tiny, 'sd_after2/')
	tiny.eval()
	with torch.no_grad():
		ev = tiny(wav, wlen)
	hyp = gen.generate(tok, ev['log_probs'][0], begin = torch.zeros(Bt), end = torch.ones(Bt), output_lengths = ev['olen'][0])
	hyp = [t[0][0]['hyp'] if len(t[0]) else '' for t in hyp]
	npz('tiny_e2e.npz', wav = wav, xlen = wlen, y = yy, ylen = yylen, eval_logits = ev['logits'][0], eval_log_probs = ev['log_probs'][0], eval_olen = ev['olen'][0], **sd0, **sd2, **steps)
	json.dump(dict(hyp = hyp, hyp_step0 = hyp_step0), open(os.path.join(HERE, 'tiny_e2e_hyp.json'), 'w'), ensure_ascii = False, indent = 1)

	# ---------------------------------------------------------------- greedy decode rules on a crafted argmax path (transcript_generators.py:27-93)
	torch.manual_seed(5)
	paths = []
	eps, sp = 37, 36
	paths.append([eps, eps, sp, 0, 0, eps, 0, 1, 1, 2] + [eps] * 12 + [3, 3, eps, sp, sp, 4, eps, eps, 5])
	paths.append([sp, eps] * 3 + [7, 7, 7, sp, eps, 8] + [eps] * 9 + [9] + [eps] * 10 + [eps, 10, sp])
	paths.append([eps] * 30)
	paths.append([11, 12, 13] + [eps] * 27)
	L = max(len(p) for p in paths)
	idx = torch.full((len(paths), L), eps)
	for i, p in enumerate(paths):
		idx[i, :len(p)] = torch.tensor(p)
	lpd = torch.full((len(paths), 38, L), -10.0).scatter_(1, idx.unsqueeze(1), 0.0)
	dlen = torch.tensor([L, L - 2, L, 10])
	hyp2 = gen.generate(tok, lpd, begin = torch.zeros(len(paths)), end = torch.ones(len(paths)), output_lengths = dlen)
	hyp2 = [t[0][0]['hyp'] if len(t[0]) else '' for t in hyp2]
	npz('decode.npz', log_probs = lpd, olen = dlen)
	json.dump(dict(hyp = hyp2), open(os.path.join(HERE, 'decode_hyp.json'), 'w'), ensure_ascii = False, indent = 1)

	# ---------------------------------------------------------------- residual (dense) wiring: small JasperNet, features in (models.py:303-313)
	torch.manual_seed(7)
	dense = models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11, 13], out_width_factors = [2, 3], dropouts = [0.2, 0.2], out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, num_subblocks = 2, check_time_dim_padded = False, temporal_mask = False)
	dense.train()
	fx = torch.randn(3, 64, 96)
	fl = torch.tensor([1.0, 0.7, 0.4])
	sd0 = sd_numpy(dense)
	out = dense(fx, fl)
	out['logits'][0].square().mean().backward()
	npz('dense_jasper.npz', x = fx, xlen = fl, logits = out['logits'][0], log_probs = out['log_probs'][0], **sd0, **{'grad/backbone.1.conv_residual.0.weight': dense.backbone[1].conv_residual[0].weight.grad, 'grad/backbone.0.conv.0.0.weight': dense.backbone[0].conv[0][0].weight.grad})

	# full-size Wav2Letter shapes (no tensors): parameter count and layer table
	w2l = models.Wav2Letter(64, [38])
	table = [dict(name = n, shape = list(p.shape)) for n, p in w2l.state_dict().items()]
	json.dump(dict(num_params = sum(p.numel() for p in w2l.parameters()), state_dict = table), open(os.path.join(HERE, 'wav2letter_layout.json'), 'w'), indent = 0)


if __name__ == '__main__':
	main()
