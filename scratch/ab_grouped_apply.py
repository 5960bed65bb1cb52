"""bn_bwd_apply_grouped (one launch, g read once) against a launch of bn_act_bwd_apply per output, same inputs: bit-identical outputs, time
per call (HIP events, 50 iterations, interleaved).  Usage: python scratch/ab_grouped_apply.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops
d = torch.device('cuda:0')
res = []
for dt in (torch.float16, ):
	for (B, T, C, n) in [(32, 626, 256, 2), (32, 626, 512, 5), (32, 626, 768, 10), (32, 1001, 768, 10), (32, 376, 640, 8)]:
		torch.manual_seed(0)
		g = ops.as_cl(torch.randn(B, C, T, device = d), dt)
		ys = [ops.as_cl(torch.randn(B, C, T, device = d), dt) for _ in range(n)]
		coefs = [torch.randn(3 * C, device = d) for _ in range(n)]
		def grouped():
			return ops.bn_bwd_apply_grouped(g, ys, coefs)
		def single():
			return [ops.bn_act_bwd_apply(g, y, c, False) for y, c in zip(ys, coefs)]
		a, b = grouped(), single()
		same = all(torch.equal(x, y) for x, y in zip(a, b))
		def timeit(fn, iters = 50):
			fn(); torch.cuda.synchronize()
			e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
			e0.record()
			for _ in range(iters):
				fn()
			e1.record(); torch.cuda.synchronize()
			return e0.elapsed_time(e1) / iters * 1e3
		tg, ts = [], []
		for _ in range(3):
			tg.append(timeit(grouped)); ts.append(timeit(single))
		unit = B * T * C * 2
		row = dict(B = B, T = T, C = C, n = n, identical = same, grouped_us = round(min(tg), 1), per_output_us = round(min(ts), 1), grouped_tbs = round((1 + 2 * n) * unit / min(tg) / 1e6, 2), per_output_tbs = round(3 * n * unit / min(ts) / 1e6, 2))
		print(row, flush = True)
		res.append(row)
if len(sys.argv) > 1:
	json.dump(dict(what = 'bn_bwd_apply_grouped (GA_ROWS = 4) vs one bn_act_bwd_apply launch per output; algorithmic TB/s', rows = res), open(sys.argv[1], 'w'), indent = 1)
