"""A/B of the fused BN-backward epilogue of the dgrad launches: matrix-pipe sums from the stored gates (form 2, default) vs the
vector-ALU form on the same gates (debug bit 256), vs the plain dgrad launch (no fused epilogue), bench shapes, one process."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); dt = torch.bfloat16
lib = _lib.load()
def timeit(fn, iters = 20):
	for _ in range(3): fn()
	torch.cuda.synchronize()
	s, e = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	s.record()
	for _ in range(iters): fn()
	e.record(); torch.cuda.synchronize()
	return s.elapsed_time(e) / iters * 1e3
B, T = 64, 751
out = {}
for (cin, cout, k, dil) in [(256, 256, 11, 1), (384, 384, 11, 1), (512, 512, 11, 1), (640, 640, 11, 1), (768, 768, 11, 1), (768, 896, 29, 2), (896, 1024, 1, 1)]:
	# dgrad of a cin -> cout conv: dy has cout channels, dx (= dz of the layer below) cin
	dy = ops.as_cl(torch.randn(B, cout, T, device = d), dt)
	w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
	_, wd = ops.pack_weight(w, dt, None)
	y = ops.as_cl(torch.randn(B, cin, T, device = d) * 4 + 2, dt)
	sc, sh = torch.rand(cin, device = d) + 0.5, torch.randn(cin, device = d)
	mean, istd = torch.randn(cin, device = d), torch.rand(cin, device = d) + 0.5
	xl = torch.ones(B, device = d)
	act = (_lib.ACT_HARDTANH, 0.0, 20.0)
	gate = torch.zeros(B * T * cin // 8, dtype = torch.uint8, device = d)
	ops.bn_act(y, sc, sh, act, xlen = xl, dropout_p = 0.2, seed = 3, offset = 5, gate = gate)
	pad = dil * (k - 1) - dil * (k // 2)
	sums = ops.ConvStats(cin, B, T, d)
	fused = lambda: ops.conv1d_dgrad_bn_reduce(dy, wd, cin, k, dil, pad, y, sc, sh, mean, istd, act, 0.2, 3, 5, xl, sums, gate = gate)
	plain = lambda: ops.conv1d(dy, wd, cin, k, 1, dil, pad)
	res = {}
	for rnd in range(2):
		for name, flags, fn in (('plain', 0, plain), ('fused_mfma', 0, fused), ('fused_valu', 256, fused)):
			lib.convasr_debug_set_conv_v2(1 | (flags << 8))
			res.setdefault(name, []).append(timeit(fn))
	lib.convasr_debug_set_conv_v2(1)
	dx2 = fused(); t2 = sums.totals().clone()
	lib.convasr_debug_set_conv_v2(1 | (256 << 8)); dx1 = fused(); t1 = sums.totals().clone(); lib.convasr_debug_set_conv_v2(1)
	rel = float((t2 - t1).abs().max() / t1.abs().max())
	best = {n: round(min(v), 1) for n, v in res.items()}
	out[f'dgrad {cout}->{cin} k{k} d{dil}'] = dict(us = best, sums_max_rel_diff = rel, dx_identical = bool(torch.equal(dx1, dx2)))
	print(f'dgrad {cout}->{cin} k{k}', best, 'sums rel diff', f'{rel:.2e}', 'dx identical', bool(torch.equal(dx1, dx2)), flush = True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r03_ab_fused_epilogue.json'), 'w'), indent = 1)
