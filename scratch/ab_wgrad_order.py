"""wgrad unit order A/B (diagnostic build `wgorder`, CONVASR_HIP_LIB=convasr_amd/libconvasr_hip.wgorder.so): order 0 shipped, 1 co-fastest,
2 3x3 (co x ci) blocks.  Mode `time`: interleaved timing of the launch pair (kernel + combine); mode `pmc`: 3 launches per (layer, order)
for a rocprofv3 --pmc FETCH_SIZE pass.  Results are identical for every order (same units, same sums) -- checked bit for bit."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else 'time'
d = torch.device('cuda:0'); dt = torch.bfloat16; torch.manual_seed(0)
lib = _lib.load()
LAYERS = [(768, 768, 11, 1), (512, 512, 11, 1), (256, 256, 11, 1), (768, 896, 29, 2)]
ORDERS = ['tap-ci-co (shipped)', 'co-ci-tap', '3x3 blocks']
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
	pad = dil * (k // 2)
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
	Tout = ops.conv_out_len(T, k, 1, dil, pad)
	dy = ops.as_cl(torch.randn(B, cout, Tout, device = d), dt)
	dw = torch.empty(k, cout, cin, device = d).permute(1, 2, 0)
	run = lambda: ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
	res, ref = {}, None
	for rnd in range(2 if mode == 'time' else 1):
		for i, name in enumerate(ORDERS):
			lib.convasr_debug_set_conv_v2(1 | ((i << 13) << 8))
			if mode == 'time':
				run(); torch.cuda.synchronize()
				ref = dw.clone() if ref is None else ref
				assert torch.equal(ref, dw), (cin, cout, name)
				res.setdefault(name, []).append(timeit(run))
			else:
				for _ in range(3): run()
				torch.cuda.synchronize()
	lib.convasr_debug_set_conv_v2(1)
	if mode == 'time':
		best = {n: min(v) for n, v in res.items()}
		out[f'{cin}->{cout} k{k} d{dil}'] = {n: dict(us = round(v, 1), vs_shipped = round(v / best[ORDERS[0]], 4)) for n, v in best.items()}
		print(f'{cin}->{cout} k{k} d{dil}', {n: f'{v:.1f} us ({v / best[ORDERS[0]]:.3f})' for n, v in best.items()}, flush = True)
if mode == 'time':
	json.dump(dict(orders = ORDERS, layers = out), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r04_ab_wgrad_order_time.json'), 'w'), indent = 1)
