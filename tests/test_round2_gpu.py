"""Round-2 parity and robustness cases on the MI355X: full-size bf16 against the fp32 oracle, BASELINE configs[4]
(JasperNetLarge, bucketed mixed-length batches, NovoGrad), side-stream weight gradients, the device-side skip gate."""
import os

import numpy as np
import pytest
import torch

from oracle import convasr_oracle as O

pytestmark = pytest.mark.gpu
FE = dict(nfft = 512, hop_length = 160)


def _cos_rel(a, b):
	a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
	return float(torch.dot(a, b) / (a.norm() * b.norm())), float((a - b).norm() / b.norm())


def test_full_wav2letter_bf16_logits_loss_and_gradients_vs_fp32_oracle():
	"""The dtype of the headline number at full model size: Wav2Letter full, 4 x 10 s, dropout 0, bf16 MFMA convolutions with fp32
	accumulation and bf16 activations between layers, against the fp32 CPU oracle on the same weights and batch.
	Tolerances (bf16 has 8 significant bits; 18 layers): logits relative L2 <= 2e-2, CTC loss <= 1e-2 relative, weight gradients of
	the first layer, the k = 29 layer and the decoder: cosine >= 0.999 and relative L2 <= 5e-2."""
	import convasr_amd as ca
	torch.manual_seed(1)
	d = torch.device('cuda:0')
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False, compute_dtype = torch.bfloat16)
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	B, secs = 4, 10
	x = torch.rand(B, 16000 * secs) * 2 - 1
	xlen = torch.linspace(0.5, 1, B)
	y = torch.randint(0, 37, (B, 1, 10 * secs))
	ylen = torch.tensor([[50], [60], [80], [100]])
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	ref = O.train_step(sd, plan, x, xlen, y, ylen, frontend = FE, lr = 0.0, momentum = 0.0, weight_decay = 0.0, max_norm = 1e30)
	model.to(d).train()
	flat = ca.train.FlatParameters(model)
	out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out['loss'] * ylen.to(d)[:, 0]).mean().backward()
	flat.finalize_grads()
	assert torch.equal(out['olen'][0].cpu(), ref['olen'])
	_, rel = _cos_rel(out['logits'][0], ref['logits'])
	assert rel <= 2e-2, ('logits rel L2', rel)
	loss_rel = float(((out['loss'].cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
	assert loss_rel <= 1e-2, ('CTC loss rel', loss_rel)
	params = dict(model.named_parameters())
	report = {}
	for k in ['backbone.0.conv.0.0.weight', 'backbone.6.conv.0.0.weight', 'backbone.3.conv.1.0.weight', 'decoder.0.weight', 'backbone.5.bn.2.weight']:
		cos, rel = _cos_rel(params[k].grad, ref['grads'][k])
		report[k] = (round(cos, 5), round(rel, 4))
	print('bf16 vs fp32 oracle (cosine, rel L2):', report, 'logits rel', rel, 'loss rel', loss_rel)
	for k, (cos, rel) in report.items():
		assert cos >= 0.999 and rel <= 5e-2, report


def test_bf16_fused_eval_greedy_strings_match_fp32_oracle_4x10s():
	"""Inference path of SURVEY 8(f1) at full size: after a few train-mode forwards (non-degenerate running statistics) the
	fused-eval bf16 model's greedy strings equal the fp32 oracle's eval strings on 4 x 10 s."""
	import convasr_amd as ca
	from convasr_amd.transcript_generators import GreedyCTCGenerator, CharTokenizerLegacy
	torch.manual_seed(2)
	d = torch.device('cuda:0')
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False)
	# a random-init network decodes to (near-)constant strings; a decoder bias spread makes the argmax frame-dependent
	with torch.no_grad():
		model.decoder[0].weight.mul_(8.0)
	B, secs = 4, 10
	x = torch.rand(B, 16000 * secs) * 2 - 1
	xlen = torch.tensor([1.0, 0.9, 0.6, 0.75])
	model.to(d).train()
	with torch.no_grad():
		for _ in range(3):
			model(x.to(d), xlen.to(d))
	sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	with torch.no_grad():
		ref = O.jasper_forward(sd, plan, x, xlen, frontend = FE, training = False)
	want = O.greedy_decode(ref['log_probs'], ref['olen'])
	model.eval()
	model.fuse_conv_bn_eval()
	tok, gen = CharTokenizerLegacy(O.CHAR_LEGACY_ALPHABET), GreedyCTCGenerator()
	got = {}
	for dt in (torch.float32, torch.bfloat16):
		model.set_compute_dtype(dt)
		with torch.no_grad():
			out = model(x.to(d), xlen.to(d))
		got[dt] = [t[0][0]['hyp'] if len(t[0]) else '' for t in gen.generate(tok, out['log_probs'][0], torch.zeros(B), torch.ones(B), output_lengths = out['olen'][0])]
		# frames whose top-2 margin is inside the compute dtype's noise may flip; the decode is compared where the oracle is decisive
		margin = ref['log_probs'].topk(2, dim = 1).values
		decisive = float(((margin[:, 0] - margin[:, 1]) > (0.05 if dt == torch.bfloat16 else 1e-3)).float().mean())
		agree = float((out['log_probs'][0].argmax(dim = 1).cpu() == ref['log_probs'].argmax(dim = 1)).float().mean())
		print(dt, 'argmax agreement', agree, 'decisive frames', decisive)
		assert agree >= decisive - 1e-3
	assert got[torch.float32] == want
	assert len(set(want)) > 1 and any(len(w) > 3 for w in want), want
	assert got[torch.bfloat16] == want, (got[torch.bfloat16], want)


def test_jaspernet_large_config4_bucketed_mixed_lengths_novograd():
	"""BASELINE configs[4]: JasperNetLarge (models.py:1407-1409, 'Jasper 10x5': dense residuals, 277 M parameters), 32 utterances
	of 5-20 s per batch from BucketingBatchSampler + collate_gpu, bf16, NovoGrad, two steps.  (The reference runs this config
	under apex amp fp16; here the reduced-precision compute type is bf16 with fp32 master weights: DESIGN.md section 7.)"""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.manual_seed(1)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.JasperNetLarge(64, [38], frontend = fe, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d).train()
	n_params = sum(p.numel() for p in model.parameters())
	assert 270e6 < n_params < 285e6, n_params
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
	ds = ca.datasets.SyntheticAudioTextDataset(256, min_duration = 5.0, max_duration = 20.0, seed = 11)
	sampler = ca.datasets.BucketingBatchSampler(ds, batch_size = 32, world_size = 1)
	sampler.set_epoch(0)
	plan = O.jasper_plan(64, [38], **O.JASPERNET_LARGE)
	seen = []

	def on_step(it, batch, res):
		meta, s, x, xlen, y, ylen = batch
		assert x.shape[0] == 32 and x.shape[1] % 128 == 0
		seen.append((x.shape[1], float(res['loss_cur']), float(res['grad_norm']), bool(res['skipped'])))

	it = ca.train.train_epoch(model, opt, ca.datasets.gpu_batches(ds, sampler, d), sampler = sampler, iteration = 0, max_iterations = 2, on_step = on_step)
	assert it == 2 and sampler.batch_idx == 2
	for T, loss, gn, skipped in seen:
		assert np.isfinite(loss) and np.isfinite(gn) and gn > 0 and not skipped, seen
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
	err = float((out['logits'][0].cpu() - ref['logits']).abs().max())
	assert err <= 1e-3 * max(scale, 1.0), ('logits', err, scale)
	loss_rel = float(((out['loss'].cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
	assert loss_rel <= 1e-4, loss_rel
	params = dict(model.named_parameters())
	names = ['backbone.10.conv_residual.0.weight', 'backbone.10.conv.4.0.weight', 'backbone.5.conv_residual.2.weight', 'backbone.0.conv.0.0.weight', 'backbone.1.bn.0.weight']
	for k in names:
		cos, rel = _cos_rel(params[k].grad, ref['grads'][k])
		assert cos >= 0.99999 and rel <= 2e-3, (k, cos, rel)


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
