"""Round 6: the split-operand ("x3") conv path -- per-op accuracy against float64, the full Wav2Letter 64 x 15 s step against the CPU oracle,
and step times of fp32 / bf16x3 / f16x3 / bf16 on one device.  Usage: python scratch/r6_split.py [ops] [full] [time]  -> gpurun_out/r06_split.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import convasr_amd as ca
from convasr_amd import ops, functional as Fn, _lib

d = torch.device('cuda:0')
what = set(sys.argv[1:]) or {'ops', 'full', 'time'}
report = {}


def rel(a, b):
	a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
	return float((a - b).norm() / b.norm())


if 'ops' in what:
	torch.manual_seed(0)
	out = {}
	for (B, Cin, Cout, T, K, dil) in [(4, 256, 384, 300, 11, 1), (2, 768, 896, 200, 29, 2), (3, 896, 1024, 257, 1, 1)]:
		pad = dil * (K // 2) if K != 29 else 29
		x = torch.randn(B, Cin, T)
		w = torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
		ref = torch.nn.functional.conv1d(x.double(), w.double(), padding = pad, dilation = dil)
		dy = torch.randn_like(ref).float() * 1e-3
		dx_ref = torch.nn.grad.conv1d_input(x.shape, w.double(), dy.double(), padding = pad, dilation = dil)
		dw_ref = torch.nn.grad.conv1d_weight(x.double(), w.shape, dy.double(), padding = pad, dilation = dil)
		xg, wg, dyg = ops.as_cl(x.to(d), torch.float32), w.to(d), ops.as_cl(dy.to(d), torch.float32)
		row = {}
		spec = Fn.ConvSpec(K, 1, dil, pad)
		for name, sp in (('bf16x3', torch.bfloat16), ('f16x3', torch.float16)):
			x3 = ops.split3(xg, sp, ops.SPLIT_INPUT)
			wf, wd = Fn.split_weight(wg, sp)
			y = ops.conv1d(x3, wf, Cout, K, 1, dil, pad, out_dtype = torch.float32)
			dy3 = ops.split3(dyg, sp, ops.SPLIT_GRAD)
			dx = ops.conv1d(dy3, wd, Cin, K, 1, dil, dil * (K - 1) - pad, out_dtype = torch.float32)
			dw = torch.empty(Cout, Cin, K, device = d)
			ops.conv1d_wgrad(ops.split3_frames(x3), ops.split3_frames(dy3), Cout, K, 1, 3 * dil, 3 * pad, dw)
			row[name] = dict(y = rel(y, ref), dx = rel(dx, dx_ref), dw = rel(dw, dw_ref))
		for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16), ('f16', torch.float16)):
			xc, dyc = ops.as_cl(xg, dt), ops.as_cl(dyg, dt)
			y = ops.conv1d(xc, ops.pack_weight(wg, dt, _lib.PACK_FWD), Cout, K, 1, dil, pad, out_dtype = torch.float32)
			dx = ops.conv1d(dyc, ops.pack_weight(wg, dt, _lib.PACK_DGRAD), Cin, K, 1, dil, dil * (K - 1) - pad, out_dtype = torch.float32)
			dw = torch.empty(Cout, Cin, K, device = d)
			ops.conv1d_wgrad(xc, dyc, Cout, K, 1, dil, pad, dw)
			row[name] = dict(y = rel(y, ref), dx = rel(dx, dx_ref), dw = rel(dw, dw_ref))
		out[f'{B}x{Cin}->{Cout}xT{T}k{K}d{dil}'] = row
		print(B, Cin, Cout, T, K, dil, json.dumps(row), flush = True)
	report['ops_rel_l2_vs_float64'] = out


def make(dt, sd0 = None, dropout = 0.0):
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = dropout, check_time_dim_padded = False, compute_dtype = dt)
	if sd0 is not None:
		assert not model.load_state_dict(sd0, strict = False).missing_keys
	return model.to(d).train()


if 'full' in what:
	from oracle import convasr_oracle as O
	B, secs = 64, 15
	g = torch.Generator().manual_seed(11)
	x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
	xlen = torch.linspace(0.5, 1, B)
	y = torch.randint(0, 37, (B, 1, 10 * secs), generator = g)
	ylen = (xlen * 8 * secs).long().clamp(min = 1).view(B, 1)
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	sd0 = O.init_state_dict(plan, seed = 1, frontend = O.frontend_config())
	torch.set_num_threads(min(os.cpu_count() or 1, 16))
	t0 = time.time()
	ref = O.train_step({k: v.clone() for k, v in sd0.items()}, plan, x, xlen, y, ylen, frontend = dict(nfft = 512, hop_length = 160), max_norm = 1e30, momentum_buffers = {})
	print('oracle step', time.time() - t0, 's', flush = True)
	names = ['decoder.0.weight', 'backbone.7.conv.0.0.weight', 'backbone.6.conv.0.0.weight', 'backbone.3.conv.1.0.weight', 'backbone.0.conv.0.0.weight', 'backbone.6.bn.0.weight']
	full = {}
	for name, dt in (('f32', torch.float32), ('bf16x3', 'bf16x3'), ('f16x3', 'f16x3'), ('f16', torch.float16), ('bf16', torch.bfloat16)):
		model = make(dt, sd0)
		out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
		lv = out['loss']
		(lv * ylen[:, 0].to(d)).mean().backward()
		params = dict(model.named_parameters())
		gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
		gn_ref = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref['grads'].values())))
		full[name] = dict(ctc_loss_rel_err_max = float(((lv.detach().float().cpu() - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max()),
			logits_rel_l2 = rel(out['logits'][0].detach().float(), ref['logits']), logits_max_abs = float((out['logits'][0].detach().float().cpu() - ref['logits']).abs().max()), logits_range = float(ref['logits'].abs().max()),
			grad_norm_rel = abs(gn - gn_ref) / gn_ref, grads_rel_l2 = {k: rel(params[k].grad, ref['grads'][k]) for k in names})
		print(name, json.dumps(full[name]), flush = True)
		del model, out, lv, params
		torch.cuda.empty_cache()
	report['full_64x15s_vs_oracle'] = full


if 'time' in what:
	B, secs = 64, 15
	torch.manual_seed(1)
	x = (torch.rand(B, 16000 * secs) * 2 - 1).to(d)
	xlen = torch.ones(B).to(d)
	y = torch.randint(0, 37, (B, 1, 10 * secs)).to(d)
	ylen = torch.full((B, 1), 10 * secs).to(d)
	times = {}
	for name, dt in (('bf16', torch.bfloat16), ('bf16x3', 'bf16x3'), ('f16x3', 'f16x3'), ('f32', torch.float32)):
		model = make(dt, dropout = 0.2)
		flat = ca.train.FlatParameters(model)
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		steps = 3 if name == 'f32' else 10
		for _ in range(2):
			ca.train.train_step(model, opt, x, xlen, y, ylen)
		torch.cuda.synchronize()
		t0 = time.time()
		for _ in range(steps):
			r = ca.train.train_step(model, opt, x, xlen, y, ylen)
		torch.cuda.synchronize()
		ms = (time.time() - t0) / steps * 1e3
		# per-family time of one more step
		timer = _lib.KernelTimer()
		_lib.timer = timer
		ca.train.train_step(model, opt, x, xlen, y, ylen)
		torch.cuda.synchronize()
		_lib.timer = None
		fam = {k: dict(launches = v['launches'], ms = round(v['total_ms'], 3), tflops = round(v['work'] / max(v['total_ms'], 1e-9) / 1e9, 1), gbps = round(v['bytes'] / max(v['total_ms'], 1e-9) / 1e6, 1)) for k, v in timer.summary().items()}
		times[name] = dict(ms_per_step = ms, audio_s_per_s = B * secs / ms * 1e3, whole_step_algorithmic_pflops = 19.18e12 / ms * 1e3 / 1e15, loss = float(r['loss']), peak_gib = torch.cuda.max_memory_allocated() / 2 ** 30, families = fam)
		print(name, json.dumps(times[name]), flush = True)
		del model, flat, opt
		torch.cuda.empty_cache()
		torch.cuda.reset_peak_memory_stats()
	report['step_time_64x15s'] = times

os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok = True)
with open(os.path.join(ROOT, 'gpurun_out', 'r06_split.json'), 'w') as f:
	json.dump(report, f, indent = 1)
print('done')
