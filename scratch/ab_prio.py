"""A/B of static wave priorities in the LDS-DMA conv kernel (diagnostic build -DCONVASR_AB_PRIO=1, CONVASR_HIP_LIB=...prio.so):
debug bit 512 = computing waves 4-7 at priority 1, 1024 = all computing waves, 2048 = the loader waves.
(The three s_setprio lines sat in front of conv_v2s.hip's main loop; they were removed after this measurement showed nothing: DESIGN.md 9.3.)"""
import os, sys
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
B, T = 64, 751
for (cin, cout, k, dil) in [(256, 256, 11, 1), (512, 512, 11, 1), (768, 768, 11, 1), (768, 896, 29, 2)]:
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)
	w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
	wp = ops.pack_weight(w, dt, _lib.PACK_FWD)
	stats = ops.ConvStats(cout, B, T, d)
	run = lambda: ops.conv1d(x, wp, cout, k, 1, dil, dil * (k // 2), stats = stats)
	res = {}
	for rnd in range(3):
		for name, flags in (('base', 0), ('young_half', 512), ('all_compute', 1024), ('loaders', 2048)):
			lib.convasr_debug_set_conv_v2(1 | (flags << 8))
			res.setdefault(name, []).append(timeit(run))
	lib.convasr_debug_set_conv_v2(1)
	print(f'{cin}->{cout} k{k}', {n: round(min(v), 1) for n, v in res.items()}, flush = True)
