"""Chunk-loop cycle shares of conv1d_wgrad_v2_kernel (diagnostic build, CONVASR_HIP_LIB=...stamps.so)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); dt = torch.bfloat16
lib = _lib.load()
for (cin, cout, k, dil) in [(768, 768, 11, 1), (256, 256, 11, 1), (512, 512, 11, 1), (768, 896, 29, 2)]:
	B, T = 64, 751
	pad = dil * k // 2
	x = ops.as_cl(torch.randn(B, cin, T, device = d), dt)
	Tout = ops.conv_out_len(T, k, 1, dil, pad)
	dy = ops.as_cl(torch.randn(B, cout, Tout, device = d), dt)
	dw = torch.empty(cout, cin, k, device = d)
	for _ in range(3): ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
	torch.cuda.synchronize()
	buf = np.zeros(256 * 8 * 8, dtype = np.uint64)
	assert lib.convasr_debug_read_wgrad_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
	s = buf.reshape(256, 8, 8).astype(np.float64)
	for grp, sl in (('waves 0-3', slice(0, 4)), ('waves 4-7', slice(4, 8))):
		m = s[:, sl, :].mean(axis = (0, 1))
		n = max(m[4], 1)
		print(f'{cin}->{cout} k{k} {grp}: chunks {n:.0f}, per chunk: work {m[0] / n:.0f} barrier_wait {m[1] / n:.0f} | loop {m[2]:.0f} epilogue {m[3]:.0f} cycles; MFMA-only floor per chunk 1024 (4 taps) ', flush = True)
