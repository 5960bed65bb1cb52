"""Where a main-loop interval of conv1d_igemm_v2s_kernel spends its cycles (diagnostic build: CONVASR_HIP_LIB=...stamps.so).
Per wave of the first 256 workgroups of ONE forward launch: prologue, [DMA issue, LDS reads + MFMA, vmcnt wait, barrier wait] summed
over the intervals, epilogue.  Shares only: the stamps serialise what the real kernel overlaps."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); dt = torch.bfloat16
lib = _lib.load()
for (cin, cout, k, dil) in [(768, 768, 11, 1), (256, 256, 11, 1), (512, 512, 11, 1), (768, 896, 29, 2)]:
	B, T = 64, 751
	x = ops.as_cl(torch.randn(B, cin, T, device = d).clamp_(0, 20), dt)  # like the network's activations: half zeros
	w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
	fwd = ops.pack_weight(w, dt, _lib.PACK_FWD)
	fused = len(sys.argv) > 1 and sys.argv[1] == 'fused'
	if fused:  # the dgrad launch with the BN-backward epilogue (dy has cout channels, dx cin)
		dy = ops.as_cl(torch.randn(B, cout, T, device = d), dt)
		_, dgr = ops.pack_weight(w, dt, None)
		yb = ops.as_cl(torch.randn(B, cin, T, device = d), dt)
		sc, sh, mean, istd = (torch.rand(cin, device = d) + 0.5 for _ in range(4))
		sums = ops.ConvStats(cin, B, T, d)
		xl = torch.ones(B, device = d)
		for _ in range(3): ops.conv1d_dgrad_bn_reduce(dy, dgr, cin, k, dil, dil * (k - 1) - dil * k // 2, yb, sc, sh, mean, istd, (_lib.ACT_HARDTANH, 0.0, 20.0), 0.2, 1, 0, xl, sums)
	else:
		import time
		t_ = time.time()
		while time.time() - t_ < 2.0:
			for _ in range(20): ops.conv1d(x, fwd, cout, k, 1, dil, dil * k // 2)
			torch.cuda.synchronize()
	torch.cuda.synchronize()
	buf = np.zeros(256 * 8 * 8, dtype = np.uint64)
	assert lib.convasr_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
	s = buf.reshape(256, 8, 8).astype(np.float64)
	if os.environ.get('STAMPS_MODE') == '2':  # build -DCONVASR_STAMPS=2: the untouched kernel with two stamps around the tile
		e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
		e0.record()
		for _ in range(20): ops.conv1d(x, fwd, cout, k, 1, dil, dil * k // 2)
		e1.record(); torch.cuda.synchronize()
		us = e0.elapsed_time(e1) / 20 * 1e3
		assert lib.convasr_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size) == 0
		s = buf.reshape(256, 8, 8).astype(np.float64)
		tile_cyc, tile_ns, start = s[:, :, 7].mean(), s[:, :, 3].mean() * 10, s[:, 0, 5] * 10
		tiles = 192 * ((cout + 127) // 128)
		print(f'{cin}->{cout} k{k}: launch {us:.1f} us, {tiles / 256:.2f} rounds; first-round tiles: {tile_cyc:.0f} cycles in {tile_ns / 1e3:.1f} us = {tile_cyc / tile_ns:.3f} GHz; rounds x tile = {-(-tiles // 256) * tile_ns / 1e3:.1f} us; start spread of the first 256 workgroups {(start.max() - start.min()) / 1e3:.1f} us; MFMA cycles per full tile {(cin // 64) * ((k + 1) // 2) * 2048 * (k / (2 * ((k + 1) // 2)))}', flush = True)
		continue
	names = ['prologue', 'epi_acc_to_lds', 'lds+mfma', 'realtime_100MHz', 'barrier_wait', 'epi_stats', 'epi_store_loop', 'tile_total']
	P = (cin // 64) * ((k + 1) // 2)
	for grp, sl in (('waves 0-3', slice(0, 4)), ('waves 4-7', slice(4, 8))):
		m = s[:, sl, :].mean(axis = (0, 1))
		print(f'{cin}->{cout} k{k} {grp}: intervals {P}, per interval: ' + f'lds+mfma {m[2] / P:.0f} barrier_wait {m[4] / P:.0f} | prologue {m[0]:.0f} | epilogue: acc->LDS {m[1]:.0f} stats {m[5]:.0f} store loop {m[6]:.0f} | tile {m[7]:.0f} cycles in {m[3] * 10:.0f} ns = {m[7] / max(m[3], 1) * 0.1:.3f} GHz in-kernel clock', flush = True)
