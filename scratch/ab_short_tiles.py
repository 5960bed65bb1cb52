"""A/B: 128-row tiles for the whole launch (debug bit 1024) vs 256-row tiles + short tails (bit 2048) on JasperNetLarge's conv shapes at
32 utterances x 376 / 626 / 876 frames, fp16, forward launches with BN statistics: which launches want which?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); dt = torch.float16; torch.manual_seed(0)
lib = _lib.load()
SHAPES = [(256, 256, 11), (256, 384, 13), (384, 384, 17), (384, 512, 17), (512, 512, 21), (512, 640, 21), (640, 640, 25), (640, 768, 25), (768, 768, 25), (768, 896, 29), (256, 768, 1), (640, 768, 1)]
B = 32
def timeit(fn, n = 10):
	for _ in range(2): fn()
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize(); e0.record()
	for _ in range(n): fn()
	e1.record(); torch.cuda.synchronize()
	return e0.elapsed_time(e1) / n * 1e3
out = {}
for T in (251, 376, 501, 626, 876):
	for (cin, cout, k) in SHAPES:
		x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
		w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
		wp = ops.pack_weight(w, dt, _lib.PACK_FWD)
		stats = ops.ConvStats(cout, B, T, d)
		run = lambda: ops.conv1d(x, wp, cout, k, 1, 1, k // 2, stats = stats)
		res, ref = {}, None
		for rnd in range(2):
			for name, bits in (('bm256', 2048), ('bm128', 1024)):
				lib.convasr_debug_set_conv_v2(1 | (bits << 8))
				y = run()
				ref = y.clone() if ref is None else ref
				assert torch.equal(ref, y), (T, cin, cout, k, name)
				res.setdefault(name, []).append(timeit(run))
		lib.convasr_debug_set_conv_v2(1)
		auto = timeit(run)
		best = {n: min(v) for n, v in res.items()}
		tiles256 = B * ((T + 255) // 256) * ((cout + 127) // 128)
		out[f'T{T} {cin}->{cout} k{k}'] = dict(tiles256 = tiles256, us256 = round(best['bm256'], 1), us128 = round(best['bm128'], 1), ratio = round(best['bm128'] / best['bm256'], 3), auto = round(auto, 1))
		print(f'T{T} {cin}->{cout} k{k} tiles256 {tiles256}: 256-row {best["bm256"]:.1f} us, 128-row {best["bm128"]:.1f} us ({best["bm128"] / best["bm256"]:.3f}), auto {auto:.1f}', flush = True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r04_ab_short_tiles.json'), 'w'), indent = 1)
