"""What do the operand bytes that miss the XCD's L2 cost?  A/B of the tile-order block shape of conv_v2s.hip (how many m x n tiles the ~32
workgroups resident on an XCD cover: 16x2 shipped; 8x4, 32x1, 11x3, 16x3, 4x8, 16x6, 6x6) in the diagnostic build `abblock`
(python -m convasr_amd.build --variant abblock -DCONVASR_AB_BLOCK=1; run with CONVASR_HIP_LIB=convasr_amd/libconvasr_hip.abblock.so).
Every shape computes the same tiles (bit-identical output), only the order -- i.e. which X / W tiles are fetched from beyond L2 how
often -- changes.  Mode `time`: interleaved timing, two rounds, best of each.  Mode `pmc`: 3 launches per (layer, shape) in a fixed
order, for a `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass (scratch/ab_block_summary.py maps dispatches back)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else 'time'
d = torch.device('cuda:0'); dt = torch.bfloat16; torch.manual_seed(0)
lib = _lib.load()
SHAPES = ['16x2', '8x4', '32x1', '11x3', '16x3', '4x8', '16x6', '6x6']
LAYERS = [(768, 768, 11, 1), (512, 512, 11, 1), (256, 256, 11, 1), (768, 896, 29, 2)]
B, T = 64, 751
def timeit(fn, n = 20):
	for _ in range(3): fn()
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize(); e0.record()
	for _ in range(n): fn()
	e1.record(); torch.cuda.synchronize()
	return e0.elapsed_time(e1) / n * 1e3
out = {}
for (cin, cout, k, dil) in LAYERS:
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
	w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
	wp = ops.pack_weight(w, dt, _lib.PACK_FWD)
	stats = ops.ConvStats(cout, B, T, d)
	pad = dil * (k // 2)
	run = lambda: ops.conv1d(x, wp, cout, k, 1, dil, pad, stats = stats)
	fl = 2.0 * B * ops.conv_out_len(T, k, 1, dil, pad) * cout * cin * k
	res, ref = {}, None
	for rnd in range(2 if mode == 'time' else 1):
		for i, name in enumerate(SHAPES):
			lib.convasr_debug_set_conv_v2(1 | ((i << 10) << 8))
			if mode == 'time':
				y = run()
				ref = y.clone() if ref is None else ref
				assert torch.equal(ref, y), (cin, cout, name)
				res.setdefault(name, []).append(timeit(run))
			else:
				for _ in range(3): run()
				torch.cuda.synchronize()
	lib.convasr_debug_set_conv_v2(1)
	if mode == 'time':
		best = {n: min(v) for n, v in res.items()}
		out[f'{cin}->{cout} k{k} d{dil}'] = {n: dict(us = round(v, 1), tflops = round(fl / v / 1e6), vs_16x2 = round(v / best['16x2'], 4)) for n, v in best.items()}
		print(f'{cin}->{cout} k{k} d{dil}', {n: f'{v:.1f} us ({v / best["16x2"]:.3f})' for n, v in best.items()}, flush = True)
if mode == 'time':
	json.dump(dict(shapes = SHAPES, layers = out), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r04_ab_block_time.json'), 'w'), indent = 1)
else:
	print('pmc order:', json.dumps(dict(layers = [f'{a}->{b} k{c} d{e}' for a, b, c, e in LAYERS], shapes = SHAPES, launches_per_cell = 3)))
