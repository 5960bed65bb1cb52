"""A/B: conv1x1.hip (debug bit 4096 forces it) vs conv_v2s.hip (bit 8192 forbids it) on one-tap shapes: JasperNetLarge's residual branches
at 32 utterances x 376 / 626 / 1001 frames (forward with bias + BN statistics, and the input gradient: a plain launch with the channel
counts swapped), Wav2Letter's 896 -> 1024 layer at 64 x 753.  Outputs must be bit-identical, BN statistics equal to fp32 rounding."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); torch.manual_seed(0)
lib = _lib.load()
def timeit(fn, n = 20):
	for _ in range(3): fn()
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize(); e0.record()
	for _ in range(n): fn()
	e1.record(); torch.cuda.synchronize()
	return e0.elapsed_time(e1) / n * 1e3
CASES = [(32, T, ci, co, torch.float16) for T in (376, 626, 1001) for (ci, co) in ((256, 256), (256, 768), (384, 512), (512, 640), (640, 768), (768, 256), (768, 640))] + [(64, 753, 896, 1024, torch.bfloat16), (64, 753, 1024, 896, torch.bfloat16), (64, 751, 256, 256, torch.bfloat16)]
out = {}
for (B, T, cin, cout, dt) in CASES:
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
	w = torch.randn(cout, cin, 1, device = d) / cin ** 0.5
	bias = torch.randn(cout, device = d)
	wp = ops.pack_weight(w, dt, _lib.PACK_FWD)
	res = {}
	ys = {}
	for with_stats in (True, False):
		stats = ops.ConvStats(cout, B, T, d) if with_stats else None
		run = lambda: ops.conv1d(x, wp, cout, 1, 1, 1, 0, bias = bias if with_stats else None, stats = stats)
		for rnd in range(2):
			for name, bits in (('v2s', 8192), ('1x1', 0), ('1x1s1', 16384)):
				lib.convasr_debug_set_conv_v2(1 | (bits << 8))
				y = run()
				key = (name, with_stats)
				if key not in ys: ys[key] = (y.clone(), stats.totals().clone() if with_stats else None)
				res.setdefault(key, []).append(timeit(run))
		lib.convasr_debug_set_conv_v2(1)
		a, b = ys[('v2s', with_stats)], ys[('1x1', with_stats)]
		assert torch.equal(a[0], b[0]) and torch.equal(a[0], ys[('1x1s1', with_stats)][0]), ('output differs', B, T, cin, cout, with_stats)
		if with_stats:
			assert float((a[1] - b[1]).abs().max()) <= 2e-6 * float(a[1].abs().max()), ('stats', B, T, cin, cout)
	best = {k: min(v) for k, v in res.items()}
	nbytes = B * T * (cin + cout) * 2
	row = dict(fwd_v2s = round(best[('v2s', True)], 1), fwd_1x1 = round(best[('1x1', True)], 1), fwd_1x1_one_stage = round(best[('1x1s1', True)], 1), plain_v2s = round(best[('v2s', False)], 1), plain_1x1 = round(best[('1x1', False)], 1), plain_1x1_one_stage = round(best[('1x1s1', False)], 1), tbps_1x1 = round(nbytes / best[('1x1', False)] / 1e6, 2))
	out[f'{B}x{T} {cin}->{cout} {str(dt)[6:]}'] = row
	print(f'{B}x{T} {cin}->{cout}', row, flush = True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r04_ab_conv1x1.json'), 'w'), indent = 1)
