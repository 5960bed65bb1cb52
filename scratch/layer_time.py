"""Un-instrumented time of single forward launches of conv1d_igemm_v2s_kernel (bf16, 64 x 751 frames): us, TFLOP/s, and cycles per tile
round at the clock rocprofv3 reports for this kernel (~2.1 GHz)."""
import sys, torch
import convasr_amd
from convasr_amd import ops, _lib
d = torch.device("cuda:0"); dt = torch.bfloat16; torch.manual_seed(0)
for (cin, cout, k, dil) in [(768, 768, 11, 1), (256, 256, 11, 1), (512, 512, 11, 1), (640, 640, 11, 1), (768, 896, 29, 2), (896, 1024, 1, 1)]:
	B, T = 64, 751
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
	w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
	fwd = ops.pack_weight(w, dt, _lib.PACK_FWD)
	stats = ops.ConvStats(cout, B, T, d)
	for _ in range(5): ops.conv1d(x, fwd, cout, k, 1, dil, dil * (k // 2), stats = stats)
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize(); e0.record()
	n = 50
	for _ in range(n): ops.conv1d(x, fwd, cout, k, 1, dil, dil * (k // 2), stats = stats)
	e1.record(); torch.cuda.synchronize()
	us = e0.elapsed_time(e1) / n * 1e3
	fl = 2.0 * B * T * cout * cin * k
	tiles = 192 * ((cout + 127) // 128)
	print(f'{cin}->{cout} k{k} d{dil}: {us:.1f} us  {fl / us / 1e6:.0f} TF/s  tiles {tiles} = {tiles / 256:.2f} rounds; MFMA cycles per tile {fl / tiles / 16384 / 4 * 16 / 1:.0f} (per SIMD); us per round {us / max(1, -(-tiles // 256)):.1f}')
