"""Un-instrumented time of single conv1d_wgrad_v2_kernel launches (+ combine), bf16, 64 x 751 frames."""
import sys, torch
import convasr_amd
from convasr_amd import ops, _lib
d = torch.device("cuda:0"); dt = torch.bfloat16; torch.manual_seed(0)
for (cin, cout, k, dil) in [(768, 768, 11, 1), (256, 256, 11, 1), (512, 512, 11, 1), (640, 640, 11, 1), (768, 896, 29, 2), (896, 1024, 1, 1)]:
	B, T = 64, 751
	pad = dil * (k // 2)
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
	dy = ops.as_cl(torch.randn(B, cout, T, device = d), dt)
	dw = torch.empty(k, cout, cin, device = d).permute(1, 2, 0)
	for _ in range(5): ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize(); e0.record()
	n = 50
	for _ in range(n): ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
	e1.record(); torch.cuda.synchronize()
	us = e0.elapsed_time(e1) / n * 1e3
	fl = 2.0 * B * T * cout * cin * k
	print(f'{cin}->{cout} k{k} d{dil}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s  checksum {float(dw.double().abs().sum()):.6e}')
