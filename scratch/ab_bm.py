"""A/B of the LDS-DMA conv kernel's tile height (256 vs 192 rows, debug bits 128 / 64) on the layers of the bench workload
(Wav2Letter full, 64 x 751 frames, bf16): forward launches (with BN statistics) and dgrad launches, each variant timed twice in
alternation in one process, outputs compared bit for bit."""
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
LAYERS = [(128, 256, 6, 1), (256, 256, 11, 1), (256, 384, 11, 1), (384, 384, 11, 1), (384, 512, 11, 1), (512, 512, 11, 1), (512, 640, 11, 1), (640, 640, 11, 1), (640, 768, 11, 1), (768, 768, 11, 1), (768, 896, 29, 2), (896, 1024, 1, 1)]
out = {}
B, T = 64, 751
for (cin, cout, k, dil) in LAYERS:
	for mode in ('fwd', 'dgrad'):
		ci, co = (cin, cout) if mode == 'fwd' else (cout, cin)
		if mode == 'dgrad' and k == 6: continue
		x = ops.as_cl(torch.randn(B, ci, T, device = d).clamp_(0, 20) if mode == 'fwd' else torch.randn(B, ci, T, device = d), dt)
		w = torch.randn(co, ci, k, device = d) / (ci * k) ** 0.5
		wp = ops.pack_weight(w, dt, _lib.PACK_FWD)
		stats = ops.ConvStats(co, B, T, d) if mode == 'fwd' else None
		pad = dil * (k // 2)
		run = lambda: ops.conv1d(x, wp, co, k, 1, dil, pad, stats = stats)
		flops = 2.0 * B * T * co * ci * k
		res, ref = {}, None
		for rnd in range(2):
			for name, flags in (('bm256', 128), ('bm192', 64), ('auto', 0)):
				lib.convasr_debug_set_conv_v2(1 | (flags << 8))
				y = run()
				ref = y.clone() if ref is None else ref
				same = bool(torch.equal(ref, y))
				us = timeit(run)
				res.setdefault(name, []).append(us)
				assert same, (cin, cout, k, name)
		lib.convasr_debug_set_conv_v2(1)
		best = {n: min(v) for n, v in res.items()}
		out[f'{mode} {ci}->{co} k{k} d{dil}'] = dict(us = {n: round(v, 1) for n, v in best.items()}, tflops = {n: round(flops / v / 1e6) for n, v in best.items()})
		print(mode, f'{ci}->{co} k{k}', {n: f'{v:.1f} us {flops / v / 1e6:.0f} TF' for n, v in best.items()}, flush = True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r03_ab_bm.json'), 'w'), indent = 1)
