"""configs[4] layer shapes (JasperNetLarge, fp16, 32 utterances): fwd / dgrad / wgrad time and TFLOP/s per distinct conv shape at
three frame counts, summed with the multiplicity each shape has in the network.  Output: gpurun_out/r4_c4_layers.json"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); dt = torch.float16; torch.manual_seed(0)
# (cin, cout, k, multiplicity): main convs, then the dense residual 1x1 convs (block j taps the inputs of blocks 1..j)
MAIN = [(256, 256, 11, 10), (256, 256, 13, 5), (256, 384, 13, 1), (384, 384, 13, 4), (384, 384, 17, 5), (384, 512, 17, 1), (512, 512, 17, 4), (512, 512, 21, 5), (512, 640, 21, 1), (640, 640, 21, 4),
	(640, 640, 25, 5), (640, 768, 25, 1), (768, 768, 25, 4), (768, 896, 29, 1), (896, 1024, 1, 1)]
cins = [256, 256, 256, 256, 384, 384, 512, 512, 640, 640]
couts = [256, 256, 256, 384, 384, 512, 512, 640, 640, 768]
RES = {}
for j, co in enumerate(couts):
	for ci in cins[:j + 1]:
		RES[(ci, co, 1)] = RES.get((ci, co, 1), 0) + 1
B = 32
def timeit(fn, n = 8):
	for _ in range(2): fn()
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize(); e0.record()
	for _ in range(n): fn()
	e1.record(); torch.cuda.synchronize()
	return e0.elapsed_time(e1) / n * 1e3
out = {}
for T in [int(a) for a in (sys.argv[1].split(',') if len(sys.argv) > 1 else '376,626,1001'.split(','))]:
	tot = dict(fwd = [0, 0], dgrad = [0, 0], wgrad = [0, 0], res_fwd = [0, 0], res_dgrad = [0, 0], res_wgrad = [0, 0])
	for (cin, cout, k, mult) in MAIN + [(a, b, c, m) for (a, b, c), m in RES.items()]:
		pad = k // 2
		x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
		w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
		fwd, dgr = ops.pack_weight(w, dt, None)
		dy = ops.as_cl(torch.randn(B, cout, T, device = d), dt)
		dw = torch.empty(k, cout, cin, device = d).permute(1, 2, 0)
		fl = 2.0 * B * T * cout * cin * k
		stats = ops.ConvStats(cout, B, T, d)
		us = dict(fwd = timeit(lambda: ops.conv1d(x, fwd, cout, k, 1, 1, pad, stats = stats)), dgrad = timeit(lambda: ops.conv1d(dy, dgr, cin, k, 1, 1, k - 1 - pad)), wgrad = timeit(lambda: ops.conv1d_wgrad(x, dy, cout, k, 1, 1, pad, dw)))
		pre = 'res_' if (k == 1 and (cin, cout) != (896, 1024)) else ''
		for n, v in us.items():
			tot[pre + n][0] += v * mult; tot[pre + n][1] += fl * mult
		out[f'T{T} {cin}->{cout} k{k} x{mult}'] = {n: [round(v, 1), round(fl / v / 1e6)] for n, v in us.items()}
		print(f'T{T} {cin:4d}->{cout:4d} k{k:2d} x{mult:2d} ' + ' | '.join(f'{n} {v:7.1f} us {fl / v / 1e6:5.0f} TF' for n, v in us.items()), flush = True)
	out[f'T{T} totals'] = {n: dict(ms = round(v[0] / 1e3, 3), tflops = round(v[1] / v[0] / 1e6)) for n, v in tot.items()}
	print(f'T{T} totals', out[f'T{T} totals'], flush = True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r4_c4_layers.json'), 'w'), indent = 1)
