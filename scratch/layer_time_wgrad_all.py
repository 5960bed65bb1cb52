"""Un-instrumented time of the weight-gradient launch pair (conv1d_wgrad_v2_kernel + combine) of every layer of the bench workload
(Wav2Letter full, bf16, 64 x 751 frames; the prologue as its stride-2 fold), TFLOP/s and the split plan's shape."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device("cuda:0"); dt = torch.bfloat16; torch.manual_seed(0)
LAYERS = [(128, 256, 6, 1), (256, 256, 11, 1), (256, 384, 11, 1), (384, 384, 11, 1), (384, 512, 11, 1), (512, 512, 11, 1), (512, 640, 11, 1), (640, 640, 11, 1), (640, 768, 11, 1), (768, 768, 11, 1), (768, 896, 29, 2), (896, 1024, 1, 1), (1024, 128, 1, 1)]
out = {}
tot_us = tot_fl = 0
for (cin, cout, k, dil) in LAYERS:
	B, T = 64, 751
	pad = dil * (k // 2) if k != 6 else 3
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
	Tout = ops.conv_out_len(T, k, 1, dil, pad)
	dy = ops.as_cl(torch.randn(B, cout, Tout, device = d), dt)
	dw = torch.empty(k, cout, cin, device = d).permute(1, 2, 0)
	for _ in range(5): ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize(); e0.record()
	n = 30
	for _ in range(n): ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
	e1.record(); torch.cuda.synchronize()
	us = e0.elapsed_time(e1) / n * 1e3
	fl = 2.0 * B * Tout * cout * cin * k
	units = ((cout + 127) // 128) * ((cin + 127) // 128) * ((k + 3) // 4)
	mult = {256: 3, 384: 2, 512: 2, 640: 2, 768: 2}.get(cin if cin == cout else -1, 1)
	tot_us += us * mult; tot_fl += fl * mult
	out[f'{cin}->{cout} k{k} d{dil}'] = dict(us = round(us, 1), tflops = round(fl / us / 1e6), units = units)
	print(f'{cin}->{cout} k{k} d{dil}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s  units {units}', flush = True)
print('sum over the step (repeats counted):', round(tot_us), 'us', round(tot_fl / tot_us / 1e6), 'TF/s')
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r03_wgrad_layers.json'), 'w'), indent = 1)
