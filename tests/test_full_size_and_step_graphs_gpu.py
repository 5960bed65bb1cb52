"""Round-5 cases on the MI355X: BASELINE configs[2]'s own configuration (Wav2Letter full, 64 x 15 s) against the CPU oracle at FULL size;
the training step replayed from HIP graphs (train.GraphedTrainStep) bit for bit against the eager step -- SGD / bf16 with dropout and a
moving learning rate, NovoGrad / fp16 under the dynamic loss scaler on a dense-residual network with the weight gradients on a side
stream; the stale "this bias gradient is zero" note."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import convasr_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FE = dict(nfft = 512, hop_length = 160)


def _dump(name, obj):
	out = os.path.join(ROOT, 'gpurun_out')
	if os.path.isdir(out):
		with open(os.path.join(out, name), 'w') as f:
			json.dump(obj, f, indent = 1)


def _cos_rel(a, b):
	a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
	return float(torch.dot(a, b) / (a.norm() * b.norm())), float((a - b).norm() / b.norm())


# ------------------------------------------------------------------------------------------------ configs[2] at its own size

def test_full_wav2letter_64x15s_fp32_and_split_operand_vs_oracle_and_16bit_losses():
	"""BASELINE configs[2] / the metric's own configuration -- Wav2Letter full, 64 utterances x 15 s, lengths linspace(0.5, 1) -- forward,
	CTC and backward on the MI355X fp32 path against the CPU oracle (models.py:282-326, train.py:745-783) at FULL size (the 4 x 10 s and
	32 x 10 s cases elsewhere are reduced batches).  Bars: per-utterance CTC loss 1e-4 relative (north_star), logits rtol 1e-3 / atol
	1e-4 x range, olen equal, gradient norm 1e-3, gradients of the decoder / an upper / the first conv and one BN gamma by cosine and
	relative L2 (the deviation of a deep gradient between two exact-fp32 implementations of this random-init network is summation-order
	noise amplified ~1.2x per layer: DESIGN section 2 -- 1.4e-2 in the first layer at 4 x 10 s); the bf16 / fp16 paths' losses of the same
	batch within 2e-3 / 3e-4 (maximum over the 64 utterances)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	B, secs = 64, 15
	g = torch.Generator().manual_seed(11)
	x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
	xlen = torch.linspace(0.5, 1, B)
	y = torch.randint(0, 37, (B, 1, 10 * secs), generator = g)
	ylen = (xlen * 8 * secs).long().clamp(min = 1).view(B, 1)  # ~8 labels per valid second: feasible for every utterance
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	sd0 = O.init_state_dict(plan, seed = 1, frontend = O.frontend_config())
	torch.set_num_threads(min(os.cpu_count() or 1, 16))
	sd = {k: v.clone() for k, v in sd0.items()}
	ref = O.train_step(sd, plan, x, xlen, y, ylen, frontend = FE, max_norm = 1e30, momentum_buffers = {})  # (max_norm: the gradients as backward left them)

	def gpu(dt):
		fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
		model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.0, check_time_dim_padded = False, compute_dtype = dt)
		assert not model.load_state_dict(sd0, strict = False).missing_keys
		return model.to(d).train()

	model = gpu(torch.float32)
	out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	loss_vec = out['loss']
	loss = (loss_vec * ylen[:, 0].to(d)).mean()
	loss.backward()
	rel_loss = float(((loss_vec.detach().cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
	scale = float(ref['logits'].abs().max())
	logit_err = float((out['logits'][0].detach().cpu() - ref['logits']).abs().max())
	assert torch.equal(out['olen'][0].cpu(), ref['olen'])
	params = dict(model.named_parameters())
	names = ['decoder.0.weight', 'backbone.7.conv.0.0.weight', 'backbone.6.conv.0.0.weight', 'backbone.3.conv.1.0.weight', 'backbone.0.conv.0.0.weight', 'backbone.6.bn.0.weight']  # head, the one-tap layer, the dilated k = 29 layer, a middle layer, the stride-2 prologue, one BN gamma
	grads = {k: _cos_rel(params[k].grad, ref['grads'][k]) for k in names}
	gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
	gn_ref = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref['grads'].values())))
	# every gradient tensor of the network, not only the six above: cosine and relative L2 against the oracle's (bias / gamma / beta vectors included)
	every = {k: _cos_rel(params[k].grad, v) for k, v in ref['grads'].items() if k in params and params[k].grad is not None and float(v.abs().max()) > 0}
	worst_rel, worst_cos = max((v[1], k) for k, v in every.items()), min((v[0], k) for k, v in every.items())
	report = dict(ctc_loss_rel_err_max = rel_loss, logits_max_abs_err = logit_err, logits_range = scale, grad_norm = gn, grad_norm_oracle = gn_ref, grads_cos_rel = grads,
		all_gradients = dict(tensors = len(every), worst_rel_l2 = worst_rel, worst_cosine = worst_cos))
	del model, out, loss, loss_vec, params
	# the split-operand path (set_compute_dtype('bf16x3'): fp32 storage, every stride-1 conv as three 16-bit MFMAs per product, csrc/split3.hip)
	# is the one that meets north_star's 1e-4 at matrix-pipe rate: held to the fp32 path's own bars on loss / logits / gradient norm, and to
	# twice the fp32 path's measured distance in the deep gradients (measured: loss 3.5e-6 / 1.4e-6, logits 4.8e-4 / 1.9e-4 of a range of 2.6,
	# first-conv gradient 2.7e-2 / 1.8e-2 where exact fp32 itself sits at 1.3e-2 from the CPU oracle: summation-order noise)
	for name in ('bf16x3', 'f16x3'):
		m3 = gpu(name)
		o3 = m3(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
		(o3['loss'] * ylen[:, 0].to(d)).mean().backward()
		p3 = dict(m3.named_parameters())
		g3 = {k: _cos_rel(p3[k].grad, ref['grads'][k]) for k in names}
		gn3 = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m3.parameters() if p.grad is not None)))
		e3 = {k: _cos_rel(p3[k].grad, v) for k, v in ref['grads'].items() if k in p3 and p3[k].grad is not None and float(v.abs().max()) > 0}
		report[name] = dict(ctc_loss_rel_err_max = float(((o3['loss'].detach().cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max()), logits_max_abs_err = float((o3['logits'][0].detach().cpu() - ref['logits']).abs().max()),
			grad_norm_rel = abs(gn3 - gn_ref) / gn_ref, grads_cos_rel = g3, olen_equal = bool(torch.equal(o3['olen'][0].cpu(), ref['olen'])),
			all_gradients = dict(tensors = len(e3), worst_rel_l2 = max((v[1], k) for k, v in e3.items()), worst_cosine = min((v[0], k) for k, v in e3.items())))
		del m3, o3, p3
	# split forward, ONE 16-bit product per gradient in the backward ('bf16x3f'): the split path's loss and logits bit for bit; its gradients and the
	# plain bf16 path's, both against the oracle, every tensor
	for name, dt in (('bf16x3f', 'bf16x3f'), ('bf16_gradients', torch.bfloat16)):
		m3 = gpu(dt)
		o3 = m3(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
		(o3['loss'].float() * ylen[:, 0].to(d)).mean().backward()
		p3 = dict(m3.named_parameters())
		gn3 = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m3.parameters() if p.grad is not None)))
		e3 = {k: _cos_rel(p3[k].grad, v) for k, v in ref['grads'].items() if k in p3 and p3[k].grad is not None and float(v.abs().max()) > 0}
		report[name] = dict(ctc_loss_rel_err_max = float(((o3['loss'].detach().float().cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max()), logits_max_abs_err = float((o3['logits'][0].detach().float().cpu() - ref['logits']).abs().max()),
			grad_norm_rel = abs(gn3 - gn_ref) / gn_ref, grads_cos_rel = {k: e3[k] for k in names}, all_gradients = dict(tensors = len(e3), worst_rel_l2 = max((v[1], k) for k, v in e3.items()), worst_cosine = min((v[0], k) for k, v in e3.items())))
		del m3, o3, p3
	for name, dt in (('bf16', torch.bfloat16), ('f16', torch.float16)):
		m16 = gpu(dt)
		with torch.no_grad():
			l16 = m16(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))['loss'].float().cpu()
		report[f'ctc_loss_rel_err_max_{name}'] = float(((l16 - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
		del m16
	_dump('r06_config2_64x15s_parity.json', report)
	for name in ('bf16x3', 'f16x3'):
		r3 = report[name]
		assert r3['olen_equal'] and r3['ctc_loss_rel_err_max'] <= 2e-5, (name, r3)  # (north_star: 1e-4)
		assert r3['logits_max_abs_err'] <= 1e-3 * scale + 1e-4 * max(scale, 1.0) and r3['grad_norm_rel'] <= 1e-3, (name, r3)
		assert r3['grads_cos_rel']['decoder.0.weight'][1] <= 1e-3 and r3['grads_cos_rel']['backbone.7.conv.0.0.weight'][1] <= 5e-3 and r3['grads_cos_rel']['backbone.6.conv.0.0.weight'][1] <= 3e-2, (name, r3)
		assert r3['grads_cos_rel']['backbone.3.conv.1.0.weight'][1] <= 6e-2 and r3['grads_cos_rel']['backbone.0.conv.0.0.weight'][1] <= 6e-2 and r3['grads_cos_rel']['backbone.0.conv.0.0.weight'][0] >= 0.998, (name, r3)
	xf, b16 = report['bf16x3f'], report['bf16_gradients']
	assert xf['ctc_loss_rel_err_max'] == report['bf16x3']['ctc_loss_rel_err_max'] and xf['logits_max_abs_err'] == report['bf16x3']['logits_max_abs_err'], (xf, report['bf16x3'])  # the same forward
	# ... and its gradients meet the full split path's own bars (measured: worst tensor 3.3e-2 / cosine 0.99945, gradient norm 3.1e-5 -- next to 3.2e-2 / 0.99950 / 2.6e-6 for
	# 'bf16x3' and 0.81 / 0.68 / 2.4e-5 for plain bf16, whose forward rounding moves activation gates and batch statistics; the one rounding of each backward operand does not)
	assert xf['all_gradients']['worst_rel_l2'][0] <= 6e-2 and xf['all_gradients']['worst_cosine'][0] >= 0.998 and xf['grad_norm_rel'] <= 1e-3, (xf, b16)
	assert xf['all_gradients']['worst_rel_l2'][0] <= b16['all_gradients']['worst_rel_l2'][0], (xf, b16)
	assert rel_loss <= 1e-4, report
	# all 56 gradient tensors (measured worst: see the dumped report): the loosest bars of the six named tensors hold for every tensor of the network
	assert len(every) >= 55 and worst_rel[0] <= 3e-2 and worst_cos[0] >= 0.9995, report['all_gradients']
	for name in ('bf16x3', 'f16x3'):
		assert report[name]['all_gradients']['worst_rel_l2'][0] <= 6e-2 and report[name]['all_gradients']['worst_cosine'][0] >= 0.998, (name, report[name]['all_gradients'])
	assert logit_err <= 1e-3 * scale + 1e-4 * max(scale, 1.0), report
	assert abs(gn - gn_ref) / gn_ref <= 1e-3, report
	assert grads['decoder.0.weight'][0] >= 0.999999 and grads['decoder.0.weight'][1] <= 1e-3, report
	assert grads['backbone.7.conv.0.0.weight'][0] >= 0.99999 and grads['backbone.7.conv.0.0.weight'][1] <= 5e-3, report
	assert grads['backbone.6.conv.0.0.weight'][0] >= 0.9999 and grads['backbone.6.conv.0.0.weight'][1] <= 1.5e-2, report
	assert grads['backbone.3.conv.1.0.weight'][0] >= 0.9995 and grads['backbone.3.conv.1.0.weight'][1] <= 3e-2, report
	assert grads['backbone.0.conv.0.0.weight'][0] >= 0.9995 and grads['backbone.0.conv.0.0.weight'][1] <= 3e-2, report
	assert grads['backbone.6.bn.0.weight'][0] >= 0.9999 and grads['backbone.6.bn.0.weight'][1] <= 1.5e-2, report
	# (measured: bf16 1.3e-3, fp16 2.0e-4 -- the MAXIMUM over 64 utterances; bench.py's parity leg quotes the maximum over its 4-utterance sample, 8.5e-4 / 6.5e-5)
	assert report['ctc_loss_rel_err_max_bf16'] <= 2e-3 and report['ctc_loss_rel_err_max_f16'] <= 3e-4, report


# ------------------------------------------------------------------------------------------------ step graphs

def _wav2letter_small(ca, d, dt, dropout):
	torch.manual_seed(3)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	return ca.models.Wav2Letter(64, [38], frontend = fe, dropout = dropout, base_width = 64, check_time_dim_padded = False, compute_dtype = dt).to(d).train()


def _dense_small(ca, d, dt, dropout):
	torch.manual_seed(4)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	return ca.models.JasperNet(64, [38], frontend = fe, base_width = 64, kernel_sizes = [11, 13, 17], out_width_factors = [2, 3, 4], dropouts = [0.2, 0.2, 0.2], out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, num_subblocks = 2, dropout = dropout, check_time_dim_padded = False, temporal_mask = False, compute_dtype = dt).to(d).train()


def _batches(d, n, shapes):
	g = torch.Generator().manual_seed(5)
	out = []
	for i in range(n):
		B, secs = shapes[i % len(shapes)]
		x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
		xlen = torch.linspace(0.6, 1, B)
		y = torch.randint(0, 37, (B, 1, 128), generator = g)
		ylen = torch.randint(10, 5 * secs, (B, 1), generator = g)
		out.append(tuple(t.to(d) for t in (x, xlen, y, ylen)))
	return out


def _run(ca, build, make_opt, batches, graphed, side_stream = False, lr_of = None, opt_level = None):
	d = torch.device('cuda:0')
	ca.functional.manual_seed(17)
	model = build()
	flat = ca.train.FlatParameters(model)
	opt = make_opt(flat)
	if opt_level is not None:
		ca.models.data_parallel_and_autocast(model, opt, opt_level = opt_level)
	ca.functional.enable_side_stream_wgrad(d, side_stream)
	stepper = ca.train.GraphedTrainStep(model, opt, max_norm = 100.0, warmup = 1, enabled = graphed)
	trace = []
	try:
		for it, (x, xlen, y, ylen) in enumerate(batches):
			if lr_of is not None:
				opt.param_groups[0]['lr'] = lr_of(it)
			r = stepper(x, xlen, y, ylen, iteration = it)
			trace.append((float(r['loss']), float(r['loss_cur']), float(r['grad_norm']), bool(r['skipped'])))
	finally:
		ca.functional.enable_side_stream_wgrad(d, False)
	torch.cuda.synchronize()
	scaler = None if flat.loss_scaler is None else flat.loss_scaler.current.tolist()
	return trace, flat.data.clone(), {k: v.clone() for k, v in model.state_dict().items() if 'running' in k or 'num_batches' in k}, stepper, scaler


def _kernel_nodes_only(stepper):
	"""Every captured step is a chain of KERNEL nodes: no memset / memcpy node (train.GraphedTrainStep._fence_transition: a memset node recorded for
	hipMemsetAsync raced with the graph's own next kernel on ROCm 7.2 when a replay followed an eagerly launched step) -- so the fence between
	eager and replayed steps is not armed."""
	assert stepper.graphs and not stepper.non_kernel_nodes
	for g in stepper.graphs.values():
		assert g['node_kinds'] is not None and set(g['node_kinds']) == {'kernel'} and g['node_kinds']['kernel'] >= 100, g['node_kinds']


def test_step_graphs_sgd_bf16_dropout_moving_lr_bitwise_equal_to_eager():
	"""20 steps over two alternating batch shapes (two graphs sharing one memory pool), dropout 0.2 (the masks of a replayed step come from
	the device-resident step key: the same as the eager step's), a learning rate that changes every step (read from device memory by the
	captured optimizer kernel): loss / grad-norm trajectory, parameters and running statistics equal bit for bit; 16 of the 20 steps replayed."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	batches = _batches(d, 20, [(6, 4), (4, 6)])
	build = lambda: _wav2letter_small(ca, d, torch.bfloat16, 0.2)
	make_opt = lambda flat: ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	lr_of = lambda it: 1e-2 * (0.9 ** it)
	eager = _run(ca, build, make_opt, batches, False, lr_of = lr_of)
	graph = _run(ca, build, make_opt, batches, True, lr_of = lr_of)
	assert graph[3].captures == 2 and graph[3].replays >= 16, (graph[3].captures, graph[3].replays, graph[3].eager_steps)
	_kernel_nodes_only(graph[3])
	assert eager[0] == graph[0], list(zip(eager[0], graph[0]))
	assert torch.equal(eager[1], graph[1])
	assert all(torch.equal(eager[2][k], graph[2][k]) for k in eager[2])
	assert eager[0][0][0] != eager[0][2][0]  # (the same batch shape at steps 0 and 2 with different data: not a frozen replay)


def test_step_graphs_novograd_fp16_dense_residuals_side_stream_bitwise_equal_to_eager():
	"""BASELINE configs[4]'s ingredients at test size: dense-residual JasperNet (1x1 residual branches with their own batch norms), fp16
	under apex's dynamic loss scaler (its start-up overflows fall into the replayed steps: the scaler's double-buffered device state is
	handed back inside the graph), NovoGrad (device-side first-step detection, EMAs double-buffered), weight gradients on the side stream
	(forked and joined inside the capture), three batch shapes."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	batches = _batches(d, 24, [(4, 5), (3, 7), (5, 4)])
	build = lambda: _dense_small(ca, d, torch.float16, 0.2)
	make_opt = lambda flat: ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
	eager = _run(ca, build, make_opt, batches, False, side_stream = True, opt_level = 'O2')
	graph = _run(ca, build, make_opt, batches, True, side_stream = True, opt_level = 'O2')
	assert graph[3].captures == 3 and graph[3].replays >= 18
	_kernel_nodes_only(graph[3])
	assert eager[0] == graph[0], list(zip(eager[0], graph[0]))
	assert torch.equal(eager[1], graph[1])
	assert all(torch.equal(eager[2][k], graph[2][k]) for k in eager[2])
	assert eager[4] == graph[4], (eager[4], graph[4])  # the scaler went through the same overflows
	print('loss scaler state after 24 steps:', eager[4])


def test_step_graphs_replayed_masks_change_from_step_to_step():
	"""The same batch replayed three times from one graph with dropout 0.5 and lr 0: the losses differ (fresh masks per replay)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	b = _batches(d, 1, [(4, 4)])[0]
	ca.functional.manual_seed(3)
	model = _wav2letter_small(ca, d, torch.bfloat16, 0.5)
	flat = ca.train.FlatParameters(model)
	opt = ca.train.SGD(flat, lr = 0.0, momentum = 0.0, weight_decay = 0.0)
	stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1)
	losses = [float(stepper(*b, iteration = it)['loss']) for it in range(6)]
	assert stepper.replays == 5 and len(set(losses)) == 6, losses


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_grouped_residual_launches_match_the_launch_per_branch_path(dt):
	"""A dense block's residual branches in grouped launches (one dispatch for the 1x1 convs + bias + BN statistics, one for their input
	gradients -- accumulated straight into the tapped outputs' gradient buffers instead of autograd's pairwise adds --, one for their
	weight gradients + combine) against a launch per branch: the forward pass is bit-identical (every problem of a grouped launch is computed
	exactly as alone), gradients agree to the rounding of the 16-bit gradient sums (a different association of the same addends)."""
	import convasr_amd as ca
	from convasr_amd import functional as Fn
	d = torch.device('cuda:0')
	x, xlen, y, ylen = _batches(d, 1, [(4, 5)])[0]
	res = {}
	for grouped in (False, True):
		prev, Fn.GROUP_RES = Fn.GROUP_RES, grouped
		try:
			model = _dense_small(ca, d, dt, 0.0)
			for blk in model.backbone:
				blk.compute_dtype = dt
			flat = ca.train.FlatParameters(model)
			out = model(x, xlen, y = y, ylen = ylen)
			out['loss'].sum().backward()
			Fn.join_side_streams()
			flat.finalize_grads()
			torch.cuda.synchronize()
			res[grouped] = (out['loss'].detach().clone(), out['logits'][0].detach().clone(), flat.grad.clone(), {n: p._convasr_grad.clone() for n, p in model.named_parameters() if hasattr(p, '_convasr_grad')})
		finally:
			Fn.GROUP_RES = prev
	assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
	worst = 0.0
	for n, ga in res[False][3].items():
		gb = res[True][3][n]
		if 'conv_residual' in n and n.endswith('bias'):
			assert float(gb.abs().max()) == 0.0 and float(ga.abs().max()) == 0.0, n
			continue
		cos, rel = _cos_rel(gb, ga)
		worst = max(worst, rel)
		assert cos >= 0.9999 and rel <= (2e-2 if dt == torch.bfloat16 else 3e-3), (n, cos, rel)
	print('grouped vs per-branch: worst relative L2 gradient difference', worst)


def test_residual_bias_gradient_is_zero_again_after_a_pass_through_eval_mode_batch_norm():
	"""train -> bn.eval() backward (which writes a real bias gradient into the arena segment) -> train must leave the
	residual conv's bias gradient at exactly zero again."""
	import convasr_amd as ca
	import torch.nn as nn
	d = torch.device('cuda:0')
	model = _dense_small(ca, d, torch.float32, 0.0)
	flat = ca.train.FlatParameters(model)
	x, xlen, y, ylen = _batches(d, 1, [(3, 4)])[0]
	blk = model.backbone[2]
	rb = blk.conv_residual[0].bias

	def fwd_bwd():
		flat.zero_grad()
		out = model(x, xlen, y = y, ylen = ylen)
		out['loss'].sum().backward()
		flat.finalize_grads()
		torch.cuda.synchronize()
	fwd_bwd()
	assert float(rb._convasr_grad.abs().max()) == 0.0
	bns = [m for m in blk.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
	for m in bns:
		m.eval()
	fwd_bwd()
	assert float(rb._convasr_grad.abs().max()) > 0.0  # through the frozen statistics the bias has a real gradient
	for m in bns:
		m.train()
	fwd_bwd()
	assert float(rb._convasr_grad.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------ grouped kernels, one by one

@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_grouped_one_tap_launches_are_bit_identical_to_one_launch_per_problem(dt):
	"""convasr_conv1x1_grouped: every problem of a grouped dispatch -- forward with bias + BN statistics, and the accumulate form the input
	gradients use -- equals the single launch (conv1x1.hip) bit for bit; 333 frames (no multiple of the 128-row tile), a 768 -> 256 problem
	(the two-stage instantiation) beside one-stage ones."""
	from convasr_amd import ops, _lib
	d = torch.device('cuda:0')
	torch.manual_seed(0)
	B, T = 3, 333
	shapes = [(128, 256), (256, 256), (384, 256), (768, 256)]
	xs = [ops.as_cl(torch.randn(B, ci, T, device = d), dt) for ci, _ in shapes]
	ws = [torch.randn(co, ci, 1, device = d) / ci ** 0.5 for ci, co in shapes]
	wps = [ops.pack_weight(w, dt, _lib.PACK_FWD) for w in ws]
	bs = [torch.randn(co, device = d) for _, co in shapes]
	sts = [ops.ConvStats(co, B, T, d) for _, co in shapes]
	ys = ops.conv1x1_grouped(xs, wps, [co for _, co in shapes], biases = bs, stats = sts)
	assert ys is not None
	for x, wp, b, st, y, (ci, co) in zip(xs, wps, bs, sts, ys, shapes):
		st1 = ops.ConvStats(co, B, T, d)
		y1 = ops.conv1d(x, wp, co, 1, 1, 1, 0, bias = b, stats = st1)
		assert torch.equal(y, y1) and st.rows == st1.rows and torch.equal(st.totals(), st1.totals()), (ci, co)
	# accumulate: out += x * w^T, the sum of the two stored 16-bit values rounded once
	base = [ops.as_cl(torch.randn(B, co, T, device = d), dt) for _, co in shapes]
	outs = [t.clone() for t in base]
	got = ops.conv1x1_grouped(xs, wps, [co for _, co in shapes], outs = outs, accumulate = [True, False, True, True])
	for i, (x, wp, (ci, co)) in enumerate(zip(xs, wps, shapes)):
		y1 = ops.conv1d(x, wp, co, 1, 1, 1, 0)
		want = (y1.float() + base[i].float()).to(dt) if i != 1 else y1
		assert torch.equal(got[i], want), (ci, co)
	assert ops.conv1x1_grouped([ops.as_cl(torch.randn(B, 96, T, device = d), dt)], [wps[0]], [256]) is None  # 96 input channels: outside the envelope, nothing launched


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_grouped_one_tap_weight_gradients_against_one_launch_per_problem(dt):
	from convasr_amd import ops
	d = torch.device('cuda:0')
	torch.manual_seed(1)
	B, T = 4, 401
	shapes = [(128, 256), (256, 384), (640, 128)]
	xs = [ops.as_cl(torch.randn(B, ci, T, device = d), dt) for ci, _ in shapes]
	dys = [ops.as_cl(torch.randn(B, co, T, device = d), dt) for _, co in shapes]
	dws = [torch.full((co, ci, 1), 7.0, device = d) for ci, co in shapes]
	zeros = [torch.full((co, ), 3.0, device = d) for _, co in shapes]
	assert ops.wgrad1x1_grouped(xs, dys, dws, zeros = [zeros[0], None, zeros[2]], accumulate = [False, True, False])
	for i, (x, dy, (ci, co)) in enumerate(zip(xs, dys, shapes)):
		ref = torch.einsum('bot,bit->oi', dy.float(), x.float()).unsqueeze(-1) + (7.0 if i == 1 else 0.0)
		err = float((dws[i] - ref).abs().max()) / float(ref.abs().max())
		assert err <= 2e-5, (ci, co, err)
	assert float(zeros[0].abs().max()) == 0.0 and float(zeros[2].abs().max()) == 0.0 and float(zeros[1].min()) == 3.0


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M', [1, 3, 11])
def test_dense_block_backward_sweep_from_gates_against_torch(dt, M):
	"""convasr_bn_bwd_reduce_many + convasr_bn_bwd_apply_grouped (pass 1 in one sweep from the stored gates, pass 2 with g read once) against
	the formulas in float64: g = gate ? dz / (1 - p) : 0 (bit-exact), sum g and sum g xhat_i per batch norm -> dgamma / dbeta and the
	coefficients, dy_i = gamma_i invstd_i (g - mean(g) - xhat_i mean(g xhat_i))."""
	from convasr_amd import ops
	d = torch.device('cuda:0')
	torch.manual_seed(M)
	B, T, C, p = 3, 211, 136, 0.25
	dz = ops.as_cl(torch.randn(B, C, T, device = d), dt)
	bits = torch.rand(B, T, C, device = d) < 0.6  # channels-last element order
	gate = (bits.view(-1, 8).to(torch.uint8) << torch.arange(8, device = d, dtype = torch.uint8)).sum(dim = 1).to(torch.uint8)
	ys = [ops.as_cl(torch.randn(B, C, T, device = d) * 2 + 1, dt) for _ in range(M)]
	means = [y.float().mean(dim = (0, 2)).contiguous() for y in ys]
	invstds = [(y.float().var(dim = (0, 2), unbiased = False) + 1e-5).rsqrt().contiguous() for y in ys]
	gammas = [torch.rand(C, device = d) + 0.5 for _ in range(M)]
	coefs = [torch.empty(3 * C, device = d) for _ in range(M)]
	dgs, dbs = [torch.full((C, ), 2.0, device = d) for _ in range(M)], [torch.zeros(C, device = d) for _ in range(M)]
	acc = [i == 0 for i in range(M)]
	g = ops.bn_bwd_reduce_many(dz, gate, p, ys, means, invstds, gammas, coefs, dgs, dbs, acc)
	thr = round(p * 65536)
	ks = torch.tensor(65536.0 / (65536 - thr), dtype = torch.float32, device = d)
	g32 = torch.where(bits.permute(0, 2, 1), dz.float() * ks, torch.zeros((), device = d))
	assert torch.equal(g, g32.to(dt))
	n = B * T
	g64 = g32.double()
	dys = ops.bn_bwd_apply_grouped(g, ys, coefs)
	for i in range(M):
		xhat = (ys[i].double() - means[i].double().view(1, C, 1)) * invstds[i].double().view(1, C, 1)
		sg, sgx = g64.sum(dim = (0, 2)), (g64 * xhat).sum(dim = (0, 2))
		tol = 2e-5 * float(sgx.abs().max() + sg.abs().max())
		assert float((dgs[i].double() - (sgx + (2.0 if acc[i] else 0.0))).abs().max()) <= tol and float((dbs[i].double() - sg).abs().max()) <= tol, i
		want = gammas[i].double().view(1, C, 1) * invstds[i].double().view(1, C, 1) * (g.double() - (sg / n).view(1, C, 1) - xhat * (sgx / n).view(1, C, 1))
		err = float((dys[i].double() - want).abs().max()) / float(want.abs().max())
		assert err <= (1.2e-2 if dt == torch.bfloat16 else 1.5e-3), (i, err)  # one 16-bit rounding of dy (and of the g the apply pass reads)


def test_step_graphs_adamw_and_train_epoch_with_a_scheduler_bitwise_equal_to_eager():
	"""train.train_epoch(..., graphs = GraphedTrainStep) with a learning-rate schedule (MultiStepLR: the rate changes between replays and reaches
	the captured AdamW kernel through device memory) and AdamW's device-side applied-step counter (double-buffered: handed back inside the
	graph) against the same epoch run eagerly: parameters, moments and the logged losses equal bit for bit."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	batches = _batches(d, 14, [(4, 4), (3, 6)])
	out = {}
	for graphed in (False, True):
		ca.functional.manual_seed(5)
		model = _wav2letter_small(ca, d, torch.bfloat16, 0.1)
		flat = ca.train.FlatParameters(model)
		opt = ca.optimizers.AdamW(flat, lr = 1e-3, betas = (0.9, 0.98), weight_decay = 1e-2)
		sched = ca.optimizers.MultiStepLR(opt, gamma = 0.5, milestones = [4, 9])
		stepper = ca.train.GraphedTrainStep(model, opt, max_norm = 10.0, warmup = 1, enabled = graphed)
		losses = []
		it = ca.train.train_epoch(model, opt, [(None, None) + b for b in batches], scheduler = sched, max_norm = 10.0, graphs = stepper, on_step = lambda i, b, r: losses.append((float(r['loss']), float(r['grad_norm']))))
		torch.cuda.synchronize()
		assert it == 14 and (not graphed or (stepper.captures == 2 and stepper.replays == 12))
		out[graphed] = (losses, flat.data.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), float(opt.applied[opt._cur, 0]), opt.param_groups[0]['lr'])
	a, b = out[False], out[True]
	assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and a[4] == b[4] == 14.0 and a[5] == b[5] == 2.5e-4


def test_learning_rate_from_device_memory_equals_the_by_value_argument():
	"""The `lr_dev` argument of the three fused optimizer steps (ABI v7): one step with the rate passed by value against one with the same rate
	read from a device float (and a WRONG by-value rate, which must be ignored)."""
	from convasr_amd import ops
	d = torch.device('cuda:0')
	torch.manual_seed(0)
	n = 4096 + 3
	p0, g = torch.randn(n, device = d), torch.randn(n, device = d)
	lr = torch.tensor([3e-3], device = d)
	a, b = p0.clone(), p0.clone()
	ba, bb = torch.zeros(n, device = d), torch.zeros(n, device = d)
	ops.sgd_step(a, g, ba, n, None, 0.0, 3e-3, 0.9, 1e-3, False, True)
	ops.sgd_step(b, g, bb, n, None, 0.0, 123.0, 0.9, 1e-3, False, True, lr_dev = lr)
	assert torch.equal(a, b) and torch.equal(ba, bb)
	a, b = p0.clone(), p0.clone()
	ma, mb, va, vb = (torch.zeros(n, device = d) for _ in range(4))
	sa, sb = torch.zeros(2, 1, device = d), torch.zeros(2, 1, device = d)
	ops.adamw_step(a, g, ma, va, n, None, 0.0, 3e-3, 0.9, 0.999, 1e-8, 1e-2, sa[0], sa[1])
	ops.adamw_step(b, g, mb, vb, n, None, 0.0, 123.0, 0.9, 0.999, 1e-8, 1e-2, sb[0], sb[1], lr_dev = lr)
	assert torch.equal(a, b) and torch.equal(ma, mb) and torch.equal(va, vb) and float(sb[1]) == 1.0


def test_frontend_normalize_signal_multiplier_against_the_oracle():
	"""models.py:499 / 570: debug_short_long_records_normalize_signal_multiplier m divides the normalised signal once more, x / ((absmax + 1e-5) m)."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.manual_seed(2)
	x = torch.rand(3, 16000) * 0.6 - 0.3
	xlen = torch.tensor([1.0, 0.7, 0.4])
	for m in (0.5, 3.0):
		fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window', debug_short_long_records_normalize_signal_multiplier = m).to(d)
		got = fe(x.to(d), xlen = xlen.to(d)).cpu()
		want = O.logmel_frontend(x, xlen, fe.window.cpu(), fe.mel.weight.cpu(), fe.mel.bias.cpu(), 512, 160, denom_multiplier = m)
		plain = O.logmel_frontend(x, xlen, fe.window.cpu(), fe.mel.weight.cpu(), fe.mel.bias.cpu(), 512, 160)
		err = float((got - want).abs().max())
		assert err <= 2e-4 * float(want.abs().max()) and float((want - plain).abs().max()) > 0.5, (m, err)


def test_step_graphs_with_frozen_blocks_bitwise_equal_to_eager():
	"""JasperNet.freeze(backbone = 2) (models.py:328-339: the first blocks' batch norms on their running statistics, their parameters out of the
	arena, dropout still applied there) under step graphs: the frozen blocks' eval-path launches -- their folded scale / shift vectors are
	recomputed inside the capture, their dropout masks follow the device-resident step key -- replay bit for bit like the eager step."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	batches = _batches(d, 10, [(4, 5), (3, 6)])
	out = {}
	for graphed in (False, True):
		ca.functional.manual_seed(21)
		model = _dense_small(ca, d, torch.bfloat16, 0.2)
		model.freeze(backbone = 2)
		flat = ca.train.FlatParameters(model)
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1, enabled = graphed)
		tr = []
		for it, b in enumerate(batches):
			r = stepper(*b, iteration = it)
			tr.append((float(r['loss']), float(r['grad_norm'])))
		torch.cuda.synchronize()
		assert not graphed or (stepper.captures == 2 and stepper.replays == 8)
		out[graphed] = (tr, flat.data.clone(), {k: v.clone() for k, v in model.state_dict().items()})
	assert out[False][0] == out[True][0] and torch.equal(out[False][1], out[True][1])
	assert all(torch.equal(v, out[True][2][k]) for k, v in out[False][2].items())
	assert len({l for l, _ in out[False][0]}) == 10


def test_logmel_other_fft_sizes_against_the_reference_and_the_oracle():
	"""models.py:516 takes nfft = 2 ** ceil(log2(window)): 256 at train.py's own defaults (8 kHz, 0.02 s), 128 / 1024 for other windows and
	rates; 80 / 128 mel channels (two per lane).  The radix-2 path of logmel_kernel against the reference's runs (frontend_nfft.npz: masked, unmasked, int16, shorter than the left
	padding), at batch size against the oracle, and LogFilterBankFrontend built with those geometries."""
	import convasr_amd as ca
	from oracle import convasr_oracle as O
	d = torch.device('cuda:0')
	g = np.load(os.path.join(GOLDEN, 'frontend_nfft.npz'))
	T_ = lambda a: torch.from_numpy(np.asarray(a))
	def close(a, b, tol, what):
		a, b = a.detach().float().cpu(), b.detach().float().cpu()
		assert a.shape == b.shape, (what, a.shape, b.shape)
		err = float((a - b).abs().max())
		assert torch.allclose(a, b, rtol = tol, atol = tol), (what, err)
	for n in sorted({k.split('/')[0] for k in g.files}):
		sr, nfft, hop, win = (int(v) for v in g[f'{n}/cfg'])
		w, mw, mb = T_(g[f'{n}/window']).to(d), T_(g[f'{n}/mel_weight']).contiguous().to(d), T_(g[f'{n}/mel_bias']).to(d)
		x, xlen = T_(g[f'{n}/x']).to(d), T_(g[f'{n}/xlen']).to(d)
		close(ca.ops.logmel(x, xlen, w, mw, mb, nfft, hop), T_(g[f'{n}/feat']), 2e-4, n + ' masked')
		close(ca.ops.logmel(x, None, w, mw, mb, nfft, hop), T_(g[f'{n}/feat_nomask']), 2e-4, n + ' unmasked')
		close(ca.ops.logmel(T_(g[f'{n}/x16']).to(d), xlen, w, mw, mb, nfft, hop), T_(g[f'{n}/feat16']), 2e-4, n + ' int16')
		close(ca.ops.logmel(T_(g[f'{n}/short']).to(d), None, w, mw, mb, nfft, hop), T_(g[f'{n}/feat_short']), 2e-4, n + ' short')
		# the module, built from the same arguments as the reference's (its buffers are the reference's to the last bit)
		nmel = mw.shape[0]
		fe = ca.models.LogFilterBankFrontend(nmel, sr, dict(sr8k_w20 = 0.02, sr8k_w10 = 0.01, sr16k_w40 = 0.04, sr44k_w20 = 0.02, sr16k_w25_m80 = 0.025, sr8k_w20_m128 = 0.02)[n], 0.01, 'hann_window').to(d)
		assert (fe.nfft, fe.hop_length, fe.win_length) == (nfft, hop, win)
		assert torch.equal(fe.mel.weight.flatten(1).cpu(), T_(g[f'{n}/mel_weight'])) and torch.equal(fe.window.cpu(), T_(g[f'{n}/window']))
		mask = ca.models.temporal_mask(x, ca.models.compute_output_lengths(x, xlen))
		close(fe(x, mask = mask), T_(g[f'{n}/feat']), 2e-4, n + ' module')
		# batch size: 16 x 12 s, ragged, against the oracle
		torch.manual_seed(3)
		B, T = 16, 12 * sr
		xs, xl = torch.rand(B, T) * 2 - 1, torch.linspace(0.4, 1, B)
		ref = O.logmel_frontend(xs, xl, w.cpu(), mw.cpu().unsqueeze(-1), mb.cpu(), nfft, hop)
		close(ca.ops.logmel(xs.to(d), xl.to(d), w, mw, mb, nfft, hop), ref, 5e-4, n + ' 16 x 12 s')


@pytest.mark.parametrize('sr, wsize, nmel, preemph, normalize', [(16000, 0.032, 40, 0.97, True), (16000, 0.02, 80, 0.97, True), (44100, 0.02, 96, 0.0, False), (16000, 0.04, 10, 0.97, True), (8000, 0.032, 64, 0.0, True), (16000, 0.02, 23, 0.97, False), (22050, 0.04, 64, 0.5, True), (8000, 0.016, 17, 0.0, False)])
def test_logmel_argument_envelope_against_the_oracle(sr, wsize, nmel, preemph, normalize):
	"""LogFilterBankFrontend's other arguments (models.py:486-526) at the FFT sizes the kernel covers: a window that fills nfft exactly (0.032 s at
	16 kHz = 512, at 8 kHz = 256), fewer mel channels than lanes (10 over 513 bins: filters wider than the kernel's 64-row LDS table, the rest read from global memory), no pre-emphasis (every frame pair takes the general load path), no
	normalisation, int16 and ragged lengths -- against the oracle's restatement of the same forward."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	fe = ca.models.LogFilterBankFrontend(nmel, sr, wsize, 0.01, 'hann_window', preemphasis = preemph, normalize_signal = normalize).to(d)
	assert fe.nfft in (128, 256, 512, 1024) and fe.win_length <= fe.nfft
	torch.manual_seed(sr + nmel)
	B, T = 5, int(2.3 * sr) + 11
	x = torch.rand(B, T) * 2 - 1
	x[3] *= 1e-3
	xlen = torch.tensor([1.0, 0.31, 0.77, 0.5, 0.02])
	for sig in (x, (x * 12000).to(torch.int16)):
		ref = O.logmel_frontend(sig, xlen, fe.window.cpu(), fe.mel.weight.cpu(), fe.mel.bias.cpu(), fe.nfft, fe.hop_length, preemphasis = preemph, normalize_signal = normalize)
		got = ca.ops.logmel(sig.to(d), xlen.to(d), fe.window, fe.mel.weight.flatten(1), fe.mel.bias, fe.nfft, fe.hop_length, preemphasis = preemph, normalize = normalize).cpu()
		assert got.shape == ref.shape == (B, nmel, 1 + T // fe.hop_length)
		# un-normalised int16 samples put the power at ~1e8..1e11: compare the logs with the same bar as everywhere else
		err = float((got - ref).abs().max())
		assert torch.allclose(got, ref, rtol = 5e-4, atol = 5e-4), (sr, wsize, nmel, str(sig.dtype), err)


def test_ctc_alignment_of_long_targets_and_the_two_kernels_agree():
	"""ctc.alignment (ctc.py:7-75) on targets longer than 511 labels -- a whole recording against its transcript, transcribe.py:176 without
	segmentation -- runs one workgroup per utterance instead of one wave: bit-equal to the reference's own run (alignment_long.npz: 700 / 1,100
	labels, 2,600 frames), equal to the one-wave kernel on every golden case both can take (debug bit 32768 routes every length to the
	workgroup kernel), the oracle's path at 2,100 labels over 6,000 frames, and a clear error beyond 8,191 labels."""
	import convasr_amd as ca
	from convasr_amd import _lib
	d = torch.device('cuda:0')
	g = np.load(os.path.join(GOLDEN, 'alignment_long.npz'))
	T_ = lambda k: torch.from_numpy(g[k])
	al = ca.ctc.alignment(T_('log_probs').to(d), T_('targets'), T_('input_lengths'), T_('target_lengths'), blank = int(g['blank']))
	assert torch.equal(al.cpu(), T_('alignment')), int((al.cpu() != T_('alignment')).sum())
	# both kernels on the short golden cases and a 64-utterance batch
	g0 = np.load(os.path.join(GOLDEN, 'alignment.npz'))
	gen = torch.Generator().manual_seed(8)
	B, C, T, S = 64, 38, 753, 150
	cases = [tuple(torch.from_numpy(g0[f'c{c}/{k}']) for k in ('log_probs', 'targets', 'input_lengths', 'target_lengths')) + (int(g0[f'c{c}/blank']), ) for c in (0, 1, 2)]
	cases.append((torch.randn(T, B, C, generator = gen).log_softmax(dim = -1), torch.randint(0, C - 1, (B, S), generator = gen), torch.randint(2 * S + 1, T + 1, (B, ), generator = gen), torch.randint(1, S + 1, (B, ), generator = gen), C - 1))
	lib = _lib.load()
	for k, (lp, tg, il, tl, blank) in enumerate(cases):
		wave = ca.ctc.alignment(lp.to(d), tg, il, tl, blank = blank)
		prev = lib.convasr_debug_set_conv_v2(1 | (32768 << 8))
		try:
			wg = ca.ctc.alignment(lp.to(d), tg, il, tl, blank = blank)
		finally:
			lib.convasr_debug_set_conv_v2(prev)
		assert torch.equal(wave, wg), (k, int((wave != wg).sum()))
	# 2,100 labels over 6,000 frames (five states per thread): against the oracle
	B, T, S = 2, 6000, 2100
	tg = torch.randint(0, C - 1, (B, S), generator = gen)
	tl, il = torch.tensor([S, 1500]), torch.tensor([T, 5000])
	logits = torch.randn(T, B, C, generator = gen)
	for b in range(B):
		pos = (torch.arange(int(tl[b])) * (int(il[b]) - 10) / int(tl[b])).long() + 3
		logits[pos, b, tg[b, :int(tl[b])]] += 6.0
	lp = logits.log_softmax(dim = -1)
	al = ca.ctc.alignment(lp.to(d), tg, il, tl, blank = C - 1).cpu()
	for b in range(B):
		a = al[b, :tl[b]]
		assert bool((a[1:] > a[:-1]).all()) and int(a[0]) >= 0 and int(a[-1]) < int(il[b]) and int(al[b, tl[b]:].abs().sum()) == 0
	threads = torch.get_num_threads()
	torch.set_num_threads(1)  # 6,000 sequential steps of tiny tensor ops: a 256-thread pool costs 13x the time of one thread here
	try:
		ref = O.ctc_alignment(lp, tg, il, tl, blank = C - 1)
	finally:
		torch.set_num_threads(threads)
	assert torch.equal(al, ref)
	with pytest.raises(_lib.ConvasrHipError, match = 'target length'):
		ca.ctc.alignment(torch.zeros(4, 1, C, device = d), torch.zeros(1, 8192, dtype = torch.int64), torch.tensor([4]), torch.tensor([1]), blank = C - 1)


def test_utterance_length_extremes_against_the_oracle():
	"""The two ends of what the reference feeds the path: --min-duration 0.1 (train.py:1005-1010: utterances of a tenth of a second, a
	handful of encoder frames, shorter than most kernels' taps and than the front end's left padding) with loss and gradients, and one
	unsegmented ten-minute recording through the whole forward (transcribe.py without --max-segment-duration: 60,001 frames, one
	utterance) -- fp32 path against the oracle, the long one also in bf16 against the fp32 logits."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	sd0 = O.init_state_dict(plan, seed = 2, frontend = O.frontend_config())

	def gpu(dt):
		fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
		model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.0, check_time_dim_padded = False, compute_dtype = dt)
		assert not model.load_state_dict(sd0, strict = False).missing_keys
		return model.to(d).train()

	threads = torch.get_num_threads()
	torch.set_num_threads(min(os.cpu_count() or 1, 16))
	try:
		# ---- 0.1 - 0.3 s
		g = torch.Generator().manual_seed(12)
		B, T = 5, 4800
		x = torch.rand(B, T, generator = g) * 2 - 1
		xlen = torch.tensor([1.0, 0.34, 0.5, 0.75, 0.4])  # 0.30 / 0.10 / 0.15 / 0.23 / 0.12 s
		y = torch.randint(0, 37, (B, 1, 3), generator = g)
		ylen = torch.tensor([[3], [1], [2], [2], [1]])
		ref = O.train_step({k: v.clone() for k, v in sd0.items()}, plan, x, xlen, y, ylen, frontend = FE, max_norm = 1e30, momentum_buffers = {})
		model = gpu(torch.float32)
		out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
		(out['loss'] * ylen[:, 0].to(d)).mean().backward()
		assert torch.equal(out['olen'][0].cpu(), ref['olen']) and int(ref['olen'].min()) <= 8
		assert bool(torch.isfinite(ref['loss_vec']).all())
		rel = float(((out['loss'].detach().cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
		scale = float(ref['logits'].abs().max())
		err = float((out['logits'][0].detach().cpu() - ref['logits']).abs().max())
		assert rel <= 1e-4 and err <= 1e-3 * scale + 1e-4 * max(scale, 1.0), (rel, err, scale)
		params = dict(model.named_parameters())
		for k in ('decoder.0.weight', 'backbone.6.conv.0.0.weight', 'backbone.0.conv.0.0.weight'):
			cos, relg = _cos_rel(params[k].grad, ref['grads'][k])
			assert cos >= 0.9995 and relg <= 3e-2, (k, cos, relg)
		del model, out, params
		# ---- one ten-minute utterance
		T = 600 * 16000
		x = torch.rand(1, T, generator = g) * 2 - 1
		xlen = torch.tensor([0.93])
		with torch.no_grad():
			ref = O.jasper_forward({k: v.clone() for k, v in sd0.items()}, plan, x, xlen, frontend = FE, training = True)
			out = gpu(torch.float32)(x.to(d), xlen.to(d))
			out16 = gpu(torch.bfloat16)(x.to(d), xlen.to(d))
		assert tuple(ref['logits'].shape) == (1, 38, 30003) and torch.equal(out['olen'][0].cpu(), ref['olen'])
		scale = float(ref['logits'].abs().max())
		err = float((out['logits'][0].cpu() - ref['logits']).abs().max())
		err16 = float((out16['logits'][0].float().cpu() - ref['logits']).abs().max())
		cos16, rel16 = _cos_rel(out16['logits'][0].float(), ref['logits'])
		_dump('r05_length_extremes.json', dict(short_ctc_rel = rel, long_logits_err = err, long_logits_range = scale, long_logits_max_err_bf16 = err16, long_logits_cos_rel_l2_bf16 = [cos16, rel16]))
		assert err <= 1e-3 * scale + 1e-4 * max(scale, 1.0), (err, scale)
		# bf16 storage through 18 layers of a random-init network: relative L2 error of the logits 0.13 here, 0.12 at 4 x 10 s -- where a CPU
		# restatement with the same storage type deviates by the same 0.1175 (tests/test_bf16_parity_gpu.py: test_full_wav2letter_16bit_whole_network_deviation_is_the_storage_types_own):
		# the storage type's own error, no larger on 30,003 frames than on 501
		assert cos16 >= 0.985 and rel16 <= 0.16 and err16 <= 0.25 * scale, (cos16, rel16, err16, scale)
	finally:
		torch.set_num_threads(threads)


FAMILY = ['Wav2LetterFlat', 'Wav2LetterResidualBig', 'Wav2LetterDenseBigLargeKernelsNoDilationNoTemporalMaskNoDropoutReLu', 'JasperNetSmallInstanceNorm', 'Wav2LetterResidualNoDilation', 'JasperNetSmallTrainableInstanceNorm']


@pytest.mark.parametrize('name', FAMILY)
@pytest.mark.parametrize('dt', ['f32', 'bf16'])
def test_named_configurations_against_the_reference(name, dt):
	"""Six of the reference's 24 named configurations (models.py:858-1404; all of them build the reference's network: tests/test_host_cpu.py) run
	by the reference itself on a (3, 64, 96) feature batch (family.npz, make_golden_r5.py): identity residuals, one residual per block over two
	sub-blocks, dense + relu + no temporal mask + per-block kernel sizes, nn.InstanceNorm1d's forward as the feature normalisation, no dilation.
	fp32: logits and three to five gradients against the reference; bf16: the same network on the MFMA path within the storage type's error."""
	import sys
	import convasr_amd as ca
	sys.path.insert(0, GOLDEN)
	from describe_model import fill_parameters
	d = torch.device('cuda:0')
	g = np.load(os.path.join(GOLDEN, 'family.npz'))
	net = getattr(ca.models, name)(64, [38], base_width = 8, dropout = 0.0, check_time_dim_padded = False, compute_dtype = torch.float32 if dt == 'f32' else torch.bfloat16)
	fill_parameters(net, 100 + FAMILY.index(name))
	net.to(d).train()
	out = net(torch.from_numpy(g['x']).to(d), torch.from_numpy(g['xlen']).to(d))
	logits = out['logits'][0]
	ref = torch.from_numpy(g[f'{name}/logits'])
	scale = float(ref.abs().max())
	logits.float().square().mean().backward()
	params = dict(net.named_parameters())
	wanted = [k[len(name) + 6:] for k in g.files if k.startswith(name + '/grad/')]
	assert len(wanted) >= 3
	if dt == 'f32':
		err = float((logits.detach().cpu() - ref).abs().max())
		assert err <= 1e-3 * scale + 1e-4, err
		for k in wanted:
			cos, rel = _cos_rel(params[k].grad, torch.from_numpy(g[f'{name}/grad/{k}']))
			assert cos >= 0.99999 and rel <= 5e-3, (k, cos, rel)
		rv = net.backbone[-1].bn[0].running_var.cpu()
		assert torch.allclose(rv, torch.from_numpy(g[f'{name}/running_var_last']), rtol = 1e-4, atol = 1e-6)
		if 'Trainable' in name:  # running statistics of the feature normalisation after this training forward, then the eval forward that uses them
			nf = net.normalize_features
			assert torch.allclose(nf.running_mean.cpu(), torch.from_numpy(g[f'{name}/features_running_mean']), rtol = 1e-5, atol = 1e-6)
			assert torch.allclose(nf.running_var.cpu(), torch.from_numpy(g[f'{name}/features_running_var']), rtol = 1e-5, atol = 1e-6) and int(nf.num_batches_tracked) == 0
			with torch.no_grad():
				ev = net.eval()(torch.from_numpy(g['x']).to(d), torch.from_numpy(g['xlen']).to(d))['logits'][0].cpu()
			want = torch.from_numpy(g[f'{name}/eval_logits'])
			assert float((ev - want).abs().max()) <= 1e-3 * float(want.abs().max()) + 1e-4
	else:
		# logits and gradients deep in a 16-bit network carry the storage type's rounding amplified layer by layer (the first conv of these 13- to 23-conv
		# networks: relative error 0.2-0.45 against fp32): the bar is the oracle's restatement of the SAME algorithm with bf16 storage at the
		# same points -- the GPU path may not deviate from the reference by more than 1.5x what that CPU restatement does, plus 0.02
		from describe_model import describe, oracle_plan
		cpu = getattr(ca.models, name)(64, [38], base_width = 8, dropout = 0.0)
		fill_parameters(cpu, 100 + FAMILY.index(name))
		sd = {k: v.clone() for k, v in cpu.state_dict().items()}
		for k in wanted:
			sd[k].requires_grad_(True)
		o = O.jasper_forward(sd, oracle_plan(describe(cpu)), torch.from_numpy(g['x']), torch.from_numpy(g['xlen']), training = True, storage = torch.bfloat16, normalize_features_temporal_mask = 'InstanceNorm' not in name, normalize_features_running = 'Trainable' in name)
		o['logits'].square().mean().backward()
		rel_gpu, rel_cpu = _cos_rel(logits.float(), ref)[1], _cos_rel(o['logits'], ref)[1]
		assert rel_gpu <= 1.5 * rel_cpu + 0.02, ('logits', rel_gpu, rel_cpu)
		_dump(f'r05_family_bf16_{name[:28]}.json', dict(logits_rel_gpu = rel_gpu, logits_rel_cpu_bf16_storage = rel_cpu))
		for k in wanted:
			want = torch.from_numpy(g[f'{name}/grad/{k}'])
			rel_gpu, rel_cpu = _cos_rel(params[k].grad, want)[1], _cos_rel(sd[k].grad, want)[1]
			assert rel_gpu <= 1.5 * rel_cpu + 0.02, (k, rel_gpu, rel_cpu)


@pytest.mark.parametrize('name', FAMILY)
def test_named_configurations_eval_and_folded_batch_norm_against_the_oracle(name):
	"""The inference path of the same five configurations (transcribe.py:44-57: model.eval(), model.fuse_conv_bn_eval()): running statistics
	populated by two train-mode forwards, then the eval forward with live batch norms and with the batch norms folded into the convs
	(models.py:141-151, residual branches included; identity residuals have nothing to fold), fp32 against the oracle's eval forward on the
	same state dict; the bf16 forward of the folded network against the fp32 one."""
	import sys
	import convasr_amd as ca
	sys.path.insert(0, GOLDEN)
	from describe_model import describe, fill_parameters, oracle_plan
	d = torch.device('cuda:0')
	g = np.load(os.path.join(GOLDEN, 'family.npz'))
	x, xlen = torch.from_numpy(g['x']), torch.from_numpy(g['xlen'])
	net = getattr(ca.models, name)(64, [38], base_width = 8, dropout = 0.0, check_time_dim_padded = False)
	fill_parameters(net, 100 + FAMILY.index(name))
	plan = oracle_plan(describe(net))
	net.to(d).train()
	with torch.no_grad():
		for scale in (1.0, 0.7):
			net(x.to(d) * scale, xlen.to(d))
		sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
		ref = O.jasper_forward(sd, plan, x, xlen, training = False, normalize_features_temporal_mask = 'InstanceNorm' not in name, normalize_features_running = 'Trainable' in name)['logits']
		rng = float(ref.abs().max())
		assert rng > 0.1  # (non-degenerate: the running statistics are populated)
		net.eval()
		live = net(x.to(d), xlen.to(d))['logits'][0].cpu()
		net.fuse_conv_bn_eval()
		assert all(isinstance(b, torch.nn.Identity) for blk in net.backbone for b in list(blk.bn) + list(blk.bn_residual))
		folded = net(x.to(d), xlen.to(d))['logits'][0].cpu()
		net.set_compute_dtype(torch.bfloat16)
		folded16 = net(x.to(d), xlen.to(d))['logits'][0].float().cpu()
	for what, got in (('live', live), ('folded', folded)):
		err = float((got - ref).abs().max())
		assert err <= 1e-3 * rng + 1e-4, (what, err, rng)
	cos, rel = _cos_rel(folded16, ref)
	assert cos >= 0.995 and rel <= 0.1, (cos, rel)


@pytest.mark.parametrize('nmel, sr, wsize', [(80, 16000, 0.025), (40, 8000, 0.02)])
def test_other_feature_counts_and_sample_rates_through_the_whole_network(nmel, sr, wsize):
	"""--num-input-features / --sample-rate / --window-size other than the benchmark's (train.py:1015-1018): the 80-dimensional filterbank at 16 kHz x
	0.025 s and train.py's default 8 kHz with 40 channels, waveform to loss and gradients through a reduced-width Wav2Letter -- fp32 against the
	oracle on the same state dict, then the bf16 MFMA path (whose stride-2 prologue then has 160 / 80 folded input channels) against the
	fp32 logits."""
	import sys
	import convasr_amd as ca
	sys.path.insert(0, GOLDEN)
	from describe_model import describe, fill_parameters, oracle_plan
	d = torch.device('cuda:0')
	def build(dt):
		fe = ca.models.LogFilterBankFrontend(nmel, sr, wsize, 0.01, 'hann_window')
		net = ca.models.Wav2Letter(nmel, [38], frontend = fe, base_width = 32, dropout = 0.0, check_time_dim_padded = False, compute_dtype = dt)
		fill_parameters(net, 7)
		return net
	net = build(torch.float32)
	sd = {k: v.clone() for k, v in net.state_dict().items()}
	plan = oracle_plan(describe(net))
	g = torch.Generator().manual_seed(nmel)
	B, T = 4, int(3.1 * sr)
	x = torch.rand(B, T, generator = g) * 2 - 1
	xlen = torch.tensor([1.0, 0.8, 0.55, 0.9])
	y = torch.randint(0, 37, (B, 1, 20), generator = g)
	ylen = torch.tensor([[20], [12], [9], [17]])
	wanted = ['backbone.0.conv.0.0.weight', 'backbone.4.conv.0.0.weight', 'decoder.0.weight']
	for k in wanted:
		sd[k].requires_grad_(True)
	ref = O.jasper_forward(sd, plan, x, xlen, y, ylen, frontend = dict(nfft = net.frontend.nfft, hop_length = net.frontend.hop_length), training = True)
	(ref['loss'] * ylen[:, 0]).mean().backward()
	net.to(d).train()
	out = net(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out['loss'] * ylen[:, 0].to(d)).mean().backward()
	rng = float(ref['logits'].detach().abs().max())
	err = float((out['logits'][0].detach().cpu() - ref['logits'].detach()).abs().max())
	rel = float(((out['loss'].detach().cpu() - ref['loss'].detach()).abs() / ref['loss'].detach().abs()).max())
	assert torch.equal(out['olen'][0].cpu(), ref['olen']) and err <= 1e-3 * rng + 1e-4 and rel <= 1e-4, (err, rng, rel)
	params = dict(net.named_parameters())
	for k in wanted:
		cos, relg = _cos_rel(params[k].grad, sd[k].grad)
		assert cos >= 0.9999 and relg <= 1.5e-2, (k, cos, relg)
	net16 = build(torch.bfloat16).to(d).train()
	out16 = net16(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out16['loss'] * ylen[:, 0].to(d)).mean().backward()
	cos, rel16 = _cos_rel(out16['logits'][0].float(), ref['logits'].detach())
	assert cos >= 0.985 and rel16 <= 0.16, (cos, rel16)  # (bf16 storage through 18 layers: 0.115 here, 0.12 on the full-width network: tests/test_bf16_parity_gpu.py)
	assert all(bool(torch.isfinite(p.grad).all()) for p in net16.parameters() if p.grad is not None)


def test_instance_norm_with_running_statistics_against_torch():
	"""nn.InstanceNorm1d(track_running_stats = True), the arithmetic MaskedInstanceNorm1d hands over to at models.py:711 (third-party: torch's
	F.instance_norm, here on the CPU): two training forwards (instance statistics out, running statistics blended in with the batch mean of
	the unbiased instance variances), then eval mode on the running statistics; 16-bit output for the MFMA path."""
	import convasr_amd as ca
	d = torch.device('cuda:0')
	eps = float(torch.finfo(torch.float16).tiny)
	ref = torch.nn.InstanceNorm1d(64, eps = eps, affine = False, track_running_stats = True)
	mine = ca.models.MaskedInstanceNorm1d(64, eps = eps, affine = False, track_running_stats = True, temporal_mask = False, legacy = False).to(d)
	g = torch.Generator().manual_seed(2)
	for step, (B, T) in enumerate([(5, 301), (3, 96)]):
		x = torch.randn(B, 64, T, generator = g) * (1 + step) + 0.3 * step
		want, got = ref.train()(x), mine.train()(x.to(d))
		assert got.shape == want.shape and float((got.cpu() - want).abs().max()) <= 2e-5
		assert torch.allclose(mine.running_mean.cpu(), ref.running_mean, rtol = 1e-5, atol = 1e-6) and torch.allclose(mine.running_var.cpu(), ref.running_var, rtol = 1e-5, atol = 1e-6)
	x = torch.randn(4, 64, 77, generator = g)
	want = ref.eval()(x)
	got = mine.eval()(x.to(d))
	assert float((got.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
	got16 = mine(x.to(d), out_dtype = torch.bfloat16, pad_time_to = 2)
	assert got16.dtype == torch.bfloat16 and got16.shape[-1] == 78 and float(got16[..., 77].abs().max()) == 0 and float((got16[..., :77].float().cpu() - want).abs().max()) <= 1e-2 * float(want.abs().max())
