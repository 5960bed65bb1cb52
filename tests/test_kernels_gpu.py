"""Kernel-level parity: every C-ABI entry point against the CPU oracle / golden vectors.  `-m gpu` needs an MI355X."""
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import convasr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
gpu = pytest.mark.gpu
T_ = lambda a: torch.as_tensor(np.asarray(a))


def dev():
	return torch.device('cuda:0')


# The 16-bit storage types of the MFMA path.  Kernel tests feed operands that are already exactly representable in the storage type,
# so the only error beyond fp32 accumulation order is ONE rounding of the output: half an ulp = 2^-8 (bf16) / 2^-11 (fp16) relative.
HALF = dict(bf16 = (torch.bfloat16, 4e-3), f16 = (torch.float16, 6e-4))
HALF_ATOL = 5e-5  # fp32 accumulation order on sums of magnitude ~1 (the fp32 tests' own bound is 2e-5)


def close(a, b, rtol, atol, what = ''):
	a, b = a.detach().double().cpu(), b.detach().double().cpu()
	assert a.shape == b.shape, (what, a.shape, b.shape)
	err = (a - b).abs()
	tol = atol + rtol * b.abs()
	assert bool((err <= tol).all()), f'{what}: max abs err {float(err.max()):.3e}, worst excess {float((err - tol).max()):.3e}'


# ------------------------------------------------------------------------------------------------ no GPU needed

def test_library_exports_every_declared_symbol():
	from convasr_amd import _lib
	header = open(os.path.join(ROOT, 'include', 'convasr_hip.h')).read()
	declared = set(re.findall(r'\b(convasr_[a-z0-9_]+)\s*\(', header))
	assert declared == set(_lib.declared_symbols()), declared ^ set(_lib.declared_symbols())
	lib = _lib.load()
	for name in declared:
		assert hasattr(lib, name), name
	assert lib.convasr_abi_version() == 10
	assert lib.convasr_conv_cout_pad(38) == 128 and lib.convasr_conv_cout_pad(256) == 256
	# every ctypes signature has as many arguments as the header's prototype
	protos = re.sub(r'/\*.*?\*/', '', header, flags = re.S)
	for m in re.finditer(r'\b(?:int|int64_t|const char\*)\s+(convasr_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;', protos, flags = re.S):
		name, args = m.group(1), m.group(2).strip()
		n = 0 if args in ('void', '') else len(args.split(','))
		assert len(_lib._SIGNATURES[name][1]) == n, (name, n, len(_lib._SIGNATURES[name][1]))


def test_argument_envelope_of_the_one_product_backward_entry_points_is_checked_before_any_launch():
	"""convasr_conv1d_wgrad_ld / convasr_pack_conv_weight_split3 refuse bad pitches / plane counts with an error code and a message (no GPU is
	touched: the checks run first, so this runs on the CPU box too)."""
	import ctypes
	from convasr_amd import _lib
	lib = _lib.load()
	p = ctypes.c_void_p(4096)  # any non-NULL value: never dereferenced
	bf16 = _lib.dtype_code(torch.bfloat16)
	args = lambda x_ld, dy_ld, dt = bf16: (p, x_ld, p, dy_ld, p, p, dt, 2, 128, 128, 64, 64, 3, 1, 1, 0, _lib.W_REFERENCE, None)
	for bad in (args(64, 128), args(3 * 128 + 4, 128), args(3 * 128, 100), args(3 * 128, 128, _lib.F32)):
		assert lib.convasr_conv1d_wgrad_ld(*bad) < 0 and b'conv1d_wgrad_ld' in lib.convasr_last_error()
	assert lib.convasr_pack_conv_weight_split3(p, _lib.W_REFERENCE, p, p, 2, bf16, 128, 128, 3, None) < 0 and b'pack_conv_weight_split3' in lib.convasr_last_error()
	# the envelope query the host asks before choosing between the in-place read and a dense copy of the plane
	q = lambda Cin, Cout, K, dil, x_ld, dt = bf16: lib.convasr_conv1d_wgrad_ld_supported(dt, 64, Cin, Cout, 753, 753, K, dil, x_ld, Cout)
	assert q(768, 768, 11, 1, 3 * 768) == 1 and q(768, 768, 29, 2, 3 * 768) == 1 and q(128, 256, 6, 1, 3 * 128) == 1
	assert q(64, 768, 11, 1, 3 * 64) == 0 and q(768, 136, 11, 1, 3 * 768) == 0 and q(768, 768, 11, 1, 3 * 768 + 4) == 0 and q(768, 768, 11, 1, 3 * 768, _lib.F32) == 0 and q(768, 768, 11, 1, 512) == 0


def test_product_path_refuses_cpu_tensors():
	from convasr_amd import ops, _lib
	with pytest.raises(_lib.ConvasrHipError):
		ops.convert(torch.zeros(1, 8, 8), torch.float32, True)


# ------------------------------------------------------------------------------------------------ layout / frontend / instnorm

@gpu
def test_convert_layout_roundtrip():
	from convasr_amd import ops
	x = torch.randn(3, 70, 131)
	for dt, tol in [(torch.float32, 0), (torch.bfloat16, 4e-3), (torch.float16, 6e-4)]:
		cl = ops.convert(x.to(dev()), dt, True)
		assert ops.is_cl(cl) and cl.shape == x.shape
		close(cl.float(), x, tol, tol, 'to channels-last')
		back = ops.convert(cl, torch.float32, False)
		assert back.is_contiguous()
		close(back, x, tol, tol, 'back')
	sl = x.to(dev())[:, 3:40, 5:100]
	close(ops.convert(sl, torch.float32, True), x[:, 3:40, 5:100], 0, 0, 'strided source')


@gpu
def test_logmel_frontend_golden():
	from convasr_amd import ops
	g = np.load(os.path.join(GOLDEN, 'frontend.npz'))
	d = dev()
	win, mw, mb = T_(g['window']).to(d), T_(g['mel_weight'])[:, :, 0].contiguous().to(d), T_(g['mel_bias']).to(d)
	x, xlen = T_(g['x']).to(d), T_(g['xlen']).to(d)
	# log-mel of white noise: the reference's own fp32 FFT differs from ours by O(1e-6) relative in power
	close(ops.logmel(x, xlen, win, mw, mb, 512, 160), T_(g['feat_masked']), 2e-4, 2e-4, 'masked')
	close(ops.logmel(x, None, win, mw, mb, 512, 160), T_(g['feat_nomask']), 2e-4, 2e-4, 'nomask')
	close(ops.logmel(T_(g['x16']).to(d), xlen, win, mw, mb, 512, 160), T_(g['feat_int16']), 2e-4, 2e-4, 'int16')
	close(ops.logmel(T_(g['short']).to(d), None, win, mw, mb, 512, 160), T_(g['feat_short']), 2e-4, 2e-4, 'short (constant pad)')


@gpu
def test_logmel_full_size_against_oracle():
	from convasr_amd import ops
	torch.manual_seed(0)
	d = dev()
	B, T = 8, 240000
	x = torch.rand(B, T) * 2 - 1
	xlen = torch.linspace(0.5, 1, B)
	win = torch.hann_window(320, periodic = True)
	mw = torch.as_tensor(O.mel_filterbank(16000, 512, 64))
	mb = torch.full((64, ), float(torch.finfo(torch.float16).tiny))
	ref = O.logmel_frontend(x, xlen, win, mw.unsqueeze(-1), mb, 512, 160)
	out = ops.logmel(x.to(d), xlen.to(d), win.to(d), mw.to(d), mb.to(d), 512, 160)
	assert out.shape == (B, 64, 1501)
	close(out, ref, 5e-4, 5e-4, 'logmel 8x15s')


@gpu
def test_instance_norm_golden():
	from convasr_amd import ops
	g = np.load(os.path.join(GOLDEN, 'instnorm.npz'))
	x, xlen = T_(g['x']).to(dev()), T_(g['xlen']).to(dev())
	close(ops.instnorm(x, xlen, 2.0 ** -14), T_(g['y_masked']), 1e-5, 1e-5, 'masked')
	close(ops.instnorm(x, None, 2.0 ** -14, channels_last = False), T_(g['y_legacy']), 1e-5, 1e-5, 'legacy')
	close(ops.instnorm(ops.as_cl(x), xlen, 2.0 ** -14, out_dtype = torch.bfloat16).float(), T_(g['y_masked']), 4e-3, 1e-5, 'bf16 out')
	close(ops.instnorm(ops.as_cl(x), xlen, 2.0 ** -14, out_dtype = torch.float16).float(), T_(g['y_masked']), 6e-4, 1e-5, 'fp16 out')


# ------------------------------------------------------------------------------------------------ conv

CONV_CASES = [
	# B, Cin, Cout, T, K, stride, dil
	(2, 64, 256, 301, 11, 2, 1),
	(3, 96, 96, 77, 11, 1, 1),
	(2, 64, 128, 140, 29, 1, 2),
	(2, 128, 38, 203, 1, 1, 1),
	(1, 256, 160, 530, 13, 1, 1),
	(2, 32, 40, 19, 3, 1, 1),
	(3, 192, 200, 700, 11, 1, 1),   # LDS-DMA kernel: 3 channel slabs, ragged T and Cout
	(2, 320, 128, 257, 1, 1, 1),    # LDS-DMA kernel: K = 1, one new X slab per step
	(1, 128, 256, 1003, 29, 1, 2),  # LDS-DMA kernel: dilated, 4 time tiles
	(5, 256, 128, 333, 11, 1, 1),   # LDS-DMA wgrad kernel: 3 tap groups (4, 4, 3), ragged chunks
	(2, 128, 384, 100, 2, 1, 1),    # LDS-DMA wgrad kernel: K = 2 (one tap pair idle)
]


def _conv_ref(x, w, bias, stride, dil):
	return O.conv_same_padding(x, w, bias, stride = stride, dilation = dil)


@gpu
@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
def test_conv1d_forward(case, dtype):
	from convasr_amd import ops, _lib
	B, Cin, Cout, T, K, stride, dil = case
	torch.manual_seed(sum(case))
	dt = torch.float32 if dtype == 'f32' else HALF[dtype][0]
	x = torch.randn(B, Cin, T)
	w = torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
	bias = torch.randn(Cout)
	if dtype != 'f32':
		x, w = x.to(dt).float(), w.to(dt).float()
	pad = dil * K // 2
	ref = _conv_ref(x, w, bias, stride, dil)
	d = dev()
	wp = ops.pack_weight(w.to(d), dt, _lib.PACK_FWD)
	y = ops.conv1d(ops.as_cl(x.to(d), dt), wp, Cout, K, stride, dil, pad, out_dtype = torch.float32 if dtype == 'f32' or Cout == 38 else dt, bias = bias.to(d))
	assert y.shape == ref.shape and ops.is_cl(y)
	if dtype == 'f32':
		close(y, ref, 1e-4, 2e-5, 'conv f32')
	elif y.dtype == torch.float32:  # fp32 output of 16-bit operands: the products are exact, only the fp32 accumulation order differs
		close(y, ref, 2e-4, 1e-4, 'conv ' + dtype + ' -> fp32')
	else:
		close(y.float(), ref, HALF[dtype][1], HALF_ATOL, 'conv ' + dtype)


@gpu
@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
@pytest.mark.parametrize('case', [(1, 256, 256, 300, 11, 1), (1, 1024, 1024, 300, 1, 1), (2, 768, 896, 150, 29, 2), (1, 512, 136, 77, 3, 1), (1, 3 * 384, 512, 300, 13, 1), (1, 128, 256, 300, 6, 1)])
def test_conv1d_split_k_forward_for_launches_of_a_few_tiles(case, dtype):
	"""convasr_conv1d_fwd_splitk (one online request is 4-16 workgroups per layer: the reduction is cut over the 64-channel input blocks, fp32 partial
	tiles, a second kernel adds them in split order and runs the epilogue) against the unsplit launch -- fp32 output: the same products, another
	association of the sum; 16-bit output: at most the last bit of a few elements -- and against float64; bias, folded scale / shift, hardtanh and
	the length mask in the epilogue; bit-identical from run to run; a batch that fills the chip is not split."""
	import ctypes
	from convasr_amd import ops, _lib
	B, Cin, Cout, T, K, dil = case
	dt = HALF[dtype][0]
	d = dev()
	torch.manual_seed(sum(case))
	x = torch.randn(B, Cin, T).to(dt).float()
	w = (torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5).to(dt).float()
	bias, scale, shift = torch.randn(Cout), torch.rand(Cout) + 0.5, torch.randn(Cout)
	xlen = torch.tensor([0.7, 1.0][:B])
	pad = dil * (K // 2)
	Tout = ops.conv_out_len(T, K, 1, dil, pad)
	lib = _lib.load()
	nb = ctypes.c_int64(0)
	splits = lib.convasr_conv1d_fwd_splitk_plan(_lib.dtype_code(dt), B, Cin, Cout, Tout, K, ctypes.byref(nb))
	assert 2 <= splits <= Cin // 64 and nb.value == splits * B * Tout * Cout * 4
	assert lib.convasr_conv1d_fwd_splitk_plan(_lib.dtype_code(dt), 64, Cin, Cout, Tout, K, ctypes.byref(nb)) == 1 and nb.value == 0  # 64 utterances fill the chip
	xg, wp = ops.as_cl(x.to(d), dt), ops.pack_weight(w.to(d), dt, _lib.PACK_FWD)
	ref = torch.nn.functional.conv1d(x.double(), w.double(), bias.double(), padding = pad, dilation = dil)
	ref_epi = torch.nn.functional.hardtanh(ref * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1), 0.0, 20.0)
	ref_epi = ref_epi * (torch.arange(Tout).view(1, 1, -1) < torch.ceil(xlen.double() * Tout).view(-1, 1, 1))
	for out_dt in (torch.float32, dt):
		plain = dict(bias = bias.to(d))
		full = dict(bias = bias.to(d), scale = scale.to(d), shift = shift.to(d), act = (_lib.ACT_HARDTANH, 0.0, 20.0), xlen = xlen.to(d))
		for kw, want in ((plain, ref), (full, ref_epi)):
			a = ops.conv1d(xg, wp, Cout, K, 1, dil, pad, out_dtype = out_dt, splitk = True, **kw)
			b = ops.conv1d(xg, wp, Cout, K, 1, dil, pad, out_dtype = out_dt, **kw)
			a2 = ops.conv1d(xg, wp, Cout, K, 1, dil, pad, out_dtype = out_dt, splitk = True, **kw)
			assert a.dtype == out_dt and a.shape == b.shape and torch.equal(a, a2)
			if out_dt == torch.float32:
				assert float((a - b).norm() / b.norm()) <= 2e-6 and float((a.double().cpu() - want).norm() / want.norm()) <= 2e-6
			else:
				ulp = 2.0 ** (-7 if dt == torch.bfloat16 else -10)
				assert float(((a.float() - b.float()).abs() / b.float().abs().clamp(min = 1e-2)).max()) <= ulp and float((a != b).float().mean()) <= 2e-2, (float((a != b).float().mean()))
				assert float((a.double().cpu() - want).norm() / want.norm()) <= HALF[dtype][1]


@gpu
@pytest.mark.parametrize('case', [(1, 256, 256, 300, 11, 1), (2, 768, 896, 150, 29, 2), (1, 512, 136, 77, 3, 1), (1, 72, 256, 300, 5, 1)])
def test_conv1d_split_k_forward_exact_fp32(case):
	"""The same for the exact-fp32 kernel (32-channel slabs, fp64 totals per workgroup rounded once into the fp32 partial, partials added in fp64): as close to
	float64 as the unsplit launch (both ~1e-7), the epilogue included; a channel count that is no multiple of the slab."""
	import ctypes
	from convasr_amd import ops, _lib
	B, Cin, Cout, T, K, dil = case
	d = dev()
	torch.manual_seed(sum(case))
	x, w = torch.randn(B, Cin, T), torch.randn(Cout, Cin, K) / (Cin * K) ** 0.5
	bias, scale, shift = torch.randn(Cout), torch.rand(Cout) + 0.5, torch.randn(Cout)
	xlen = torch.tensor([0.7, 1.0][:B])
	pad = dil * (K // 2)
	Tout = ops.conv_out_len(T, K, 1, dil, pad)
	nb = ctypes.c_int64(0)
	splits = _lib.load().convasr_conv1d_fwd_splitk_plan(_lib.F32, B, Cin, Cout, Tout, K, ctypes.byref(nb))
	assert 2 <= splits <= (Cin + 31) // 32 and nb.value == splits * B * Tout * Cout * 4
	xg, wp = ops.as_cl(x.to(d), torch.float32), ops.pack_weight(w.to(d), torch.float32, _lib.PACK_FWD)
	ref = torch.nn.functional.conv1d(x.double(), w.double(), bias.double(), padding = pad, dilation = dil)
	ref_epi = torch.nn.functional.hardtanh(ref * scale.double().view(1, -1, 1) + shift.double().view(1, -1, 1), 0.0, 20.0)
	ref_epi = ref_epi * (torch.arange(Tout).view(1, 1, -1) < torch.ceil(xlen.double() * Tout).view(-1, 1, 1))
	full = dict(bias = bias.to(d), scale = scale.to(d), shift = shift.to(d), act = (_lib.ACT_HARDTANH, 0.0, 20.0), xlen = xlen.to(d))
	for kw, want in ((dict(bias = bias.to(d)), ref), (full, ref_epi)):
		a = ops.conv1d(xg, wp, Cout, K, 1, dil, pad, splitk = True, **kw)
		b = ops.conv1d(xg, wp, Cout, K, 1, dil, pad, **kw)
		ea, eb = float((a.double().cpu() - want).norm() / want.norm()), float((b.double().cpu() - want).norm() / want.norm())
		assert torch.equal(a, ops.conv1d(xg, wp, Cout, K, 1, dil, pad, splitk = True, **kw)) and ea <= max(1.5 * eb, 3e-7) and float((a - b).norm() / b.norm()) <= 5e-7, (ea, eb)


@gpu
@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[5] == 1 and c[1] % 64 == 0])
def test_conv1d_lds_dma_kernel_is_bit_identical_to_register_staged_kernel(case):
	from convasr_amd import ops, _lib
	B, Cin, Cout, T, K, stride, dil = case
	torch.manual_seed(sum(case) + 5)
	d = dev()
	x = ops.as_cl(torch.randn(B, Cin, T, device = d), torch.bfloat16)
	w = torch.randn(Cout, Cin, K, device = d) / (Cin * K) ** 0.5
	wp = ops.pack_weight(w, torch.bfloat16, _lib.PACK_FWD)
	xlen = torch.linspace(0.4, 1, B, device = d)
	outs = []
	for use_v2 in (1, 0):
		prev = _lib.load().convasr_debug_set_conv_v2(use_v2)
		stats = torch.zeros(2 * Cout, dtype = torch.float64, device = d)
		y = ops.conv1d(x, wp, Cout, K, 1, dil, dil * K // 2, stats = stats, act = (_lib.ACT_RELU, 0.0, 0.0), xlen = xlen)
		_lib.load().convasr_debug_set_conv_v2(prev)
		outs.append((y, stats))
	assert torch.equal(outs[0][0], outs[1][0])
	close(outs[0][1], outs[1][1], 1e-6, 1e-4, 'stats')  # fp32 per-lane partial sums are grouped differently by the two MFMA shapes


@gpu
def test_conv1d_epilogue_stats_scale_act_mask():
	from convasr_amd import ops, _lib
	torch.manual_seed(3)
	B, Cin, Cout, T, K = 3, 64, 200, 150, 11
	x, w = torch.randn(B, Cin, T), torch.randn(Cout, Cin, K) / 26
	scale, shift = torch.rand(Cout) + 0.5, torch.randn(Cout)
	xlen = torch.tensor([1.0, 0.61, 0.3])
	raw = _conv_ref(x, w, None, 1, 1)
	lengths = O.compute_output_lengths(T, xlen)
	ref = F.hardtanh(raw * scale[None, :, None] + shift[None, :, None], 0, 20) * O.temporal_mask(T, lengths).unsqueeze(1)
	d = dev()
	stats = torch.zeros(2 * Cout, dtype = torch.float64, device = d)
	y = ops.conv1d(ops.as_cl(x.to(d)), ops.pack_weight(w.to(d), torch.float32, _lib.PACK_FWD), Cout, K, 1, 1, 5, stats = stats, scale = scale.to(d), shift = shift.to(d), act = (_lib.ACT_HARDTANH, 0.0, 20.0), xlen = xlen.to(d))
	close(y, ref, 1e-4, 2e-5, 'fused epilogue')
	close(stats[:Cout], raw.double().sum(dim = (0, 2)), 1e-5, 1e-4, 'sum')
	close(stats[Cout:], (raw.double() ** 2).sum(dim = (0, 2)), 1e-5, 1e-4, 'sumsq')


@gpu
@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[5] == 1])
@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
def test_conv1d_dgrad(case, dtype):
	from convasr_amd import ops, _lib
	B, Cin, Cout, T, K, stride, dil = case
	torch.manual_seed(sum(case) + 1)
	dt = torch.float32 if dtype == 'f32' else HALF[dtype][0]
	w = torch.randn(Cout, Cin, K) / (Cout * K) ** 0.5
	pad = dil * K // 2
	Tout = ops.conv_out_len(T, K, 1, dil, pad)
	dy = torch.randn(B, Cout, Tout)
	if dtype != 'f32':
		dy, w = dy.to(dt).float(), w.to(dt).float()
	x = torch.zeros(B, Cin, T, requires_grad = True)
	_conv_ref(x, w, None, 1, dil).backward(dy)
	d = dev()
	wp = ops.pack_weight(w.to(d), dt, _lib.PACK_DGRAD)
	dx = ops.conv1d(ops.as_cl(dy.to(d), dt), wp, Cin, K, 1, dil, dil * (K - 1) - pad)
	assert dx.shape == x.shape
	if dtype == 'f32':
		close(dx, x.grad, 1e-4, 2e-5, 'dgrad f32')
	else:
		close(dx.float(), x.grad, HALF[dtype][1], HALF_ATOL, 'dgrad ' + dtype)


@gpu
@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16', 'f16'])
def test_conv1d_wgrad(case, dtype):
	from convasr_amd import ops
	B, Cin, Cout, T, K, stride, dil = case
	torch.manual_seed(sum(case) + 2)
	dt = torch.float32 if dtype == 'f32' else HALF[dtype][0]
	pad = dil * K // 2
	Tout = ops.conv_out_len(T, K, stride, dil, pad)
	x, dy = torch.randn(B, Cin, T), torch.randn(B, Cout, Tout)
	if dtype != 'f32':
		x, dy = x.to(dt).float(), dy.to(dt).float()
	w = torch.zeros(Cout, Cin, K, requires_grad = True)
	b = torch.zeros(Cout, requires_grad = True)
	_conv_ref(x, w, b, stride, dil).backward(dy)
	d = dev()
	dw = torch.full((Cout, Cin, K), 7.0, device = d)
	db = torch.full((Cout, ), 7.0, device = d)
	ops.conv1d_wgrad(ops.as_cl(x.to(d), dt), ops.as_cl(dy.to(d), dt), Cout, K, stride, dil, pad, dw, dbias = db)
	tol = dict(rtol = 1e-4, atol = 1e-3) if dtype == 'f32' else dict(rtol = 2e-4, atol = 2e-3)  # 16-bit: the gradient is fp32 and the operands multiply exactly; only the (b, t) summation order differs (was 2e-2 / 5e-2 sqrt(N) / 10)
	close(dw, w.grad, what = 'wgrad', **tol)
	close(db, b.grad, what = 'dbias', **tol)
	ops.conv1d_wgrad(ops.as_cl(x.to(d), dt), ops.as_cl(dy.to(d), dt), Cout, K, stride, dil, pad, dw, dbias = db, accumulate = True)
	close(dw, 2 * w.grad, what = 'wgrad accumulate', rtol = tol['rtol'], atol = 2 * tol['atol'])


@gpu
@pytest.mark.parametrize('shape', [(256, 128, 11), (384, 256, 3), (128, 128, 29), (40, 64, 5)])
def test_tap_major_weights_pack_wgrad_and_optimizer_mirror(shape):
	"""CONVASR_W_KMAJOR (the training arena's weight / gradient layout, include/convasr_hip.h): packing from a tap-major master gives
	the same packed operands as packing the torch-contiguous tensor, a tap-major weight gradient is the reference-layout one
	permuted -- bit for bit (same slabs, same summation order) -- and the optimizer's bf16 mirror equals a cast of the parameters."""
	from convasr_amd import ops, _lib
	Cout, Cin, K = shape
	d = dev()
	torch.manual_seed(Cout + K)
	w = torch.randn(Cout, Cin, K, device = d)
	wk = torch.empty(K, Cout, Cin, device = d).permute(1, 2, 0)
	wk.copy_(w)
	assert ops.weight_layout(w) == _lib.W_REFERENCE and ops.weight_layout(wk) == _lib.W_KMAJOR and ops.weight_layout(w.permute(0, 2, 1)) is None
	for dt in (torch.float32, torch.bfloat16):
		f0, g0 = ops.pack_weight(w, dt, None)
		f1, g1 = ops.pack_weight(wk, dt, None)
		assert torch.equal(f0, f1) and torch.equal(g0, g1), dt
		# only the dgrad copy rebuilt from a forward copy that is already current
		g2 = torch.zeros_like(g1)
		ops.pack_weight(wk, dt, None, out = (f1, g2), fwd_is_current = True)
		assert torch.equal(g2, g0)
	B, T = 3, 300
	x = ops.as_cl(torch.randn(B, Cin, T, device = d), torch.bfloat16)
	dy = ops.as_cl(torch.randn(B, Cout, T, device = d), torch.bfloat16)
	dw = torch.empty(Cout, Cin, K, device = d)
	dwk = torch.full((K, Cout, Cin), 3.0, device = d).permute(1, 2, 0)
	ops.conv1d_wgrad(x, dy, Cout, K, 1, 1, K // 2, dw)
	ops.conv1d_wgrad(x, dy, Cout, K, 1, 1, K // 2, dwk)
	assert torch.equal(dw, dwk.contiguous())
	ops.conv1d_wgrad(x, dy, Cout, K, 1, 1, K // 2, dwk, accumulate = True)
	close(dwk.contiguous(), 2 * dw, 1e-6, 1e-6, 'tap-major accumulate')
	# optimizer mirror
	n = 4096 + 64
	p_, g_, buf = torch.randn(n, device = d), torch.randn(n, device = d), torch.zeros(n, device = d)
	mirror = torch.zeros(n, dtype = torch.bfloat16, device = d)
	ref = p_.clone()
	ops.sgd_step(p_, g_, buf, n, ops.sumsq(g_), 100.0, 1e-2, 0.9, 1e-3, False, True, p16 = mirror)
	assert not torch.equal(p_, ref) and torch.equal(mirror, p_.to(torch.bfloat16))
	keep = mirror.clone()
	ops.sgd_step(p_, g_, buf, n, ops.sumsq(g_), 100.0, 1e-2, 0.9, 1e-3, False, False, loss_gate = torch.tensor([float('inf')], device = d), p16 = mirror)
	assert torch.equal(mirror, keep)


# ------------------------------------------------------------------------------------------------ batch norm + activation

@gpu
@pytest.mark.parametrize('C', [64, 384, 1024, 2560])
@pytest.mark.parametrize('nonlin', [('hardtanh', 0, 20), ('relu', ), ('leaky_relu', 0.01)])
def test_bn_act_forward_backward(C, nonlin):
	from convasr_amd import ops
	torch.manual_seed(C)
	B, T = 3, 57
	y = (torch.randn(B, C, T) * 3 + 1).requires_grad_(True)
	gamma, beta = (torch.rand(C) + 0.5).requires_grad_(True), torch.randn(C).requires_grad_(True)
	rm, rv = torch.zeros(C), torch.ones(C)
	xlen = torch.tensor([1.0, 0.5, 0.77])
	mask = O.temporal_mask(T, O.compute_output_lengths(T, xlen)).unsqueeze(1)
	z = O.activation(F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5), nonlin) * mask
	dz = torch.randn_like(z)
	z.backward(dz)
	d = dev()
	ycl = ops.as_cl(y.detach().to(d))
	stats = torch.stack([y.detach().double().sum(dim = (0, 2)), (y.detach().double() ** 2).sum(dim = (0, 2))]).reshape(-1).to(d)
	rm_d, rv_d = torch.zeros(C, device = d), torch.ones(C, device = d)
	mean, invstd, scale, shift = ops.bn_finalize(stats, B * T, gamma.detach().to(d), beta.detach().to(d), rm_d, rv_d, 0.1, 1e-5)
	close(rm_d, rm, 1e-5, 1e-6, 'running_mean')
	close(rv_d, rv, 1e-5, 1e-6, 'running_var')
	act = ops.act_args(nonlin)
	zz = ops.bn_act(ycl, scale, shift, act, xlen = xlen.to(d))
	close(zz, z, 1e-4, 1e-4, 'bn_act fwd')
	sums = torch.zeros(2 * C, dtype = torch.float64, device = d)
	g = ops.bn_act_bwd_reduce(ops.as_cl(dz.to(d)), ycl, scale, shift, mean, invstd, act, xlen = xlen.to(d), sums = sums)
	dgamma, dbeta = torch.empty(C, device = d), torch.empty(C, device = d)
	dy = ops.bn_bwd_apply(g, ycl, gamma.detach().to(d), mean, invstd, sums, dgamma, dbeta)
	close(dgamma, gamma.grad, 1e-4, 1e-3, 'dgamma')
	close(dbeta, beta.grad, 1e-4, 1e-3, 'dbeta')
	close(dy, y.grad, 1e-3, 1e-4, 'dy')
	# fused form: pass 1 reduces only (no g), emits coefficients + parameter gradients; pass 2 recomputes g from dz
	coef = torch.empty(3 * C, device = d)
	dgamma2, dbeta2 = torch.full((C, ), 5.0, device = d), torch.full((C, ), -5.0, device = d)
	assert ops.bn_act_bwd_reduce(ops.as_cl(dz.to(d)), ycl, scale, shift, mean, invstd, act, xlen = xlen.to(d), write_g = False, gamma = gamma.detach().to(d), coef = coef, dgamma = dgamma2, dbeta = dbeta2, accumulate = True) is None
	close(dgamma2 - 5.0, gamma.grad, 1e-4, 1e-3, 'dgamma (fused, accumulate)')
	close(dbeta2 + 5.0, beta.grad, 1e-4, 1e-3, 'dbeta (fused, accumulate)')
	dy2 = ops.bn_act_bwd_apply(ops.as_cl(dz.to(d)), ycl, coef, True, scale, shift, act, xlen = xlen.to(d))
	close(dy2, y.grad, 1e-3, 1e-4, 'dy (fused)')


@gpu
def test_bn_act_dropout_statistics_and_backward_consistency():
	from convasr_amd import ops
	d = dev()
	B, C, T = 4, 256, 500
	y = ops.as_cl(torch.rand(B, C, T, device = d) + 1)
	act = ops.act_args(('relu', ))
	z = ops.bn_act(y, None, None, act, dropout_p = 0.2, seed = 1234, offset = 77)
	keep = (z != 0).float().mean().item()
	assert abs(keep - 0.8) < 0.01
	close(z[z != 0], (y / 0.8)[z != 0], 1e-5, 1e-6, 'scaled by 1/(1-p)')  # p is quantised to 16 bits: 13107/65536
	z2 = ops.bn_act(y, None, None, act, dropout_p = 0.2, seed = 1234, offset = 77)
	assert torch.equal(z, z2)
	g = ops.bn_act_bwd_reduce(torch.ones_like(y), y, None, None, None, None, act, dropout_p = 0.2, seed = 1234, offset = 77)
	assert torch.equal(g != 0, z != 0)


# ------------------------------------------------------------------------------------------------ head

@gpu
def test_weighted_mean_entropy_and_normalize_signal_golden():
	"""models.py:660-686 helpers against vectors produced by the reference (tests/golden/make_golden_r2.py)."""
	import convasr_amd as ca
	g = np.load(os.path.join(GOLDEN, 'helpers.npz'))
	d = dev()
	lp, olen = T_(g['log_probs']).to(d), T_(g['olen']).to(d)
	close(ca.models.weighted_mean_entropy(lp, olen), T_(g['wme_len']), 2e-5, 1e-6, 'weighted_mean_entropy')
	close(ca.models.weighted_mean_entropy(lp), T_(g['wme_all']), 2e-5, 1e-6, 'weighted_mean_entropy (no lengths)')
	close(ca.models.weighted_mean_entropy(lp, olen, eps_id = 3), T_(g['wme_id3']), 2e-5, 1e-6, 'weighted_mean_entropy (eps_id 3)')
	close(ca.models.entropy(lp, olen, dim = 1), T_(g['ent_len']), 2e-5, 1e-6, 'entropy')
	sig = T_(g['signal']).to(d)
	close(ca.models.normalize_signal(sig), T_(g['signal_norm']), 3e-7, 1e-9, 'normalize_signal')  # reciprocal multiply vs division: 1-2 ulp
	close(ca.models.normalize_signal(sig, denom_multiplier = 2.5), T_(g['signal_norm_mult']), 3e-7, 1e-9, 'normalize_signal (multiplier)')
	with pytest.raises(ca._lib.ConvasrHipError):
		ca.models.weighted_mean_entropy(lp, olen, dim = 0)


@gpu
def test_log_softmax_entropy_argmax():
	from convasr_amd import ops
	torch.manual_seed(0)
	B, C, T = 5, 38, 211
	logits = torch.randn(B, C, T) * 3
	olen = torch.tensor([211, 100, 1, 57, 210])
	lp_ref = F.log_softmax(logits, dim = 1)
	d = dev()
	lp = ops.log_softmax(ops.as_cl(logits.to(d)))
	close(lp, lp_ref, 1e-5, 1e-5, 'log_softmax')
	close(ops.entropy(lp, olen), O.entropy(lp_ref, olen), 1e-4, 1e-5, 'entropy')
	close(ops.entropy(lp), O.entropy(lp_ref), 1e-4, 1e-5, 'entropy (no lengths)')
	assert torch.equal(ops.argmax(lp).cpu(), lp_ref.argmax(dim = 1))
	g = torch.randn(B, C, T)
	lg = logits.clone().requires_grad_(True)
	F.log_softmax(lg, dim = 1).backward(g)
	close(ops.log_softmax_bwd(g.to(d), lp), lg.grad, 1e-4, 1e-5, 'log_softmax bwd')


@gpu
def test_ctc_golden():
	from convasr_amd import ops
	g = np.load(os.path.join(GOLDEN, 'ctc.npz'))
	d = dev()
	lp = ops.as_cl(T_(g['log_probs']).to(d))
	nll, grad = ops.ctc_loss(lp, T_(g['targets']), T_(g['olen']), T_(g['ylen']), 37)
	ref = T_(g['loss'])
	assert torch.isinf(nll[5]) and nll[5] > 0
	close(nll[:5], ref[:5], 1e-5, 1e-5, 'nll')
	w = T_(g['grad_weights'])
	close(ops.scale_rows(grad, w.to(d))[:5], T_(g['grad'])[:5], 1e-4, 1e-4, 'grad')
	assert float(grad[1, :, 40:].abs().max()) == 0.0
	assert torch.isfinite(grad).all()  # infeasible utterance: zeros (ATen leaves NaN; the step is skipped either way)


@gpu
@pytest.mark.parametrize('shape', [(8, 38, 753, 150), (3, 38, 120, 1), (2, 129, 300, 40), (2, 38, 64, 31), (2, 38, 1001, 100), (2, 38, 1200, 100)])
def test_ctc_against_oracle(shape):
	from convasr_amd import ops
	B, C, T, S = shape
	torch.manual_seed(T)
	lp = torch.randn(B, C, T).log_softmax(dim = 1)
	y = torch.randint(0, C - 1, (B, S))
	y[0, : S // 2] = y[0, 0]
	olen = torch.randint(max(T // 2, 2 * S + 1), T + 1, (B, ))
	olen[0] = T
	ylen = torch.randint(max(S // 2, 1), S + 1, (B, ))
	ylen[-1] = S
	# the oracle runs in float64: at T = 753 ATen's own fp32 lattice is only good to ~1.2e-3 in the gradient (measured against
	# float64, scratch/ctc_prec.py), the renormalised fp32 lattice of the HIP kernel to ~1e-4
	lpr = lp.double().requires_grad_(True)
	ref = O.ctc_loss(lpr, y, olen, ylen)
	ref[torch.isfinite(ref)].sum().backward()
	nll, grad = ops.ctc_loss(ops.as_cl(lp.to(dev())), y, olen, ylen, C - 1)
	fin = torch.isfinite(ref)
	assert torch.equal(torch.isfinite(nll).cpu(), fin)
	close(nll.cpu()[fin], ref.detach().float()[fin], 1e-5, 1e-4, 'nll')  # BASELINE bar: 1e-4 relative
	close(grad.cpu()[fin], lpr.grad.float()[fin], 1e-4, 2e-4, 'grad vs float64 oracle')
	lp32 = lp.clone().requires_grad_(True)
	ref32 = O.ctc_loss(lp32, y, olen, ylen)
	ref32[torch.isfinite(ref32)].sum().backward()
	close(grad.cpu()[fin], lp32.grad[fin], 1e-3, 3e-3 if T > 500 else 1e-4, 'grad vs float32 oracle')


@gpu
@pytest.mark.parametrize('S,T', [(64, 300), (128, 400), (200, 600), (300, 753), (383, 900), (447, 1000), (511, 1100), (575, 1300), (639, 1400), (703, 1600), (767, 1700), (831, 1900),
	(895, 2000), (959, 2200), (1023, 2400)])
def test_ctc_every_split_of_the_states_over_the_two_waves(S, T):
	"""csrc/ctc.hip gives the two waves of a sweep NPH and NPL (blank, label) pairs per lane: (1,1) (2,1) (2,2) (3,2) (3,3) (4,3) (4,4) for
	up to 511 labels, (5,4) ... (8,8) for up to 1,023 (round 5: F.ctc_loss takes any length; 20 s of fast speech stays below 511).  One case per split, target lengths chosen so that the last two states of the extended target sit on the two sides of
	the waves' common edge (S = 64, 128) or deep in the upper wave; T = 900 .. 1100 runs with the log-probs left in global memory."""
	from convasr_amd import ops
	B, C = 2, 38
	torch.manual_seed(S)
	lp = (torch.randn(B, C, T) * 2).log_softmax(dim = 1)
	y = torch.randint(0, C - 1, (B, S))
	y[1, : S // 3] = y[1, 0]   # repeats: no s-2 transitions there, and 2 S + 1 - (S // 3 - 1) more frames needed
	olen = torch.tensor([T, T - 7])
	ylen = torch.tensor([S, S])
	lpr = lp.double().requires_grad_(True)
	ref = O.ctc_loss(lpr, y, olen, ylen)
	assert torch.isfinite(ref).all()
	ref.sum().backward()
	nll, grad = ops.ctc_loss(ops.as_cl(lp.to(dev())), y, olen, ylen, C - 1)
	close(nll.cpu(), ref.detach().float(), 1e-5, 1e-4, 'nll')
	# the fp32 lattice's error in the gradient grows with the number of frames it is carried over (renormalised every 8 frames: ~1e-4 at
	# T = 753, 2.6e-4 at 1,700, 4.4e-4 at 2,200-2,400, where ATen's unnormalised fp32 lattice is at 1e-3 already at 753): the bar of the <= 1,100-frame cases, scaled
	close(grad.cpu(), lpr.grad.float(), 1e-4, 2e-4 * max(1.0, T / 1100) ** 1.5, 'grad vs float64 oracle')
	if T > 1100:  # ... and never worse than the reference's own arithmetic (F.ctc_loss in fp32) is against float64
		lp32 = lp.clone().requires_grad_(True)
		O.ctc_loss(lp32, y, olen, ylen).sum().backward()
		err_ref, err_hip = float((lp32.grad.double() - lpr.grad).abs().max()), float((grad.cpu().double() - lpr.grad).abs().max())
		assert err_hip <= err_ref, (err_hip, err_ref)


@gpu
def test_ctc_infinite_and_nan_log_probs():
	"""-inf log-probs are staged as the kernel's finite sentinel: a forbidden class off the target changes nothing, a forbidden class ON
	the target makes the utterance infeasible (+inf, zero gradient -- F.ctc_loss: +inf).  A NaN must not hang the wave that polls the
	other wave's edge states (their "not written yet" marker is a NaN pattern): the call returns and the loss is not finite."""
	from convasr_amd import ops
	B, C, T, S = 3, 38, 200, 150   # (2, 1) split: both waves of each sweep hold real states
	torch.manual_seed(1)
	lp = torch.randn(B, C, T).log_softmax(dim = 1)
	y = torch.randint(0, C - 2, (B, S))   # class C - 2 never occurs in a target
	lp[0, C - 2] = -float('inf')
	lp[1, int(y[1, 140]), :] = -float('inf')
	olen, ylen = torch.full((B, ), T), torch.full((B, ), S)
	# consecutive equal labels need a blank between them: T = 200 frames hold 150 labels only if at most 50 repeats
	for b in range(B):
		for i in range(1, S):
			if y[b, i] == y[b, i - 1]:
				y[b, i] = (y[b, i] + 1) % (C - 2)
	lp[1, :, :] = torch.randn(C, T).log_softmax(dim = 0)
	lp[1, int(y[1, 140]), :] = -float('inf')
	lpr = lp.double().requires_grad_(True)
	ref = O.ctc_loss(lpr, y, olen, ylen)
	assert torch.isfinite(ref[0]) and torch.isinf(ref[1]) and torch.isfinite(ref[2])
	(ref[0] + ref[2]).backward()
	nll, grad = ops.ctc_loss(ops.as_cl(lp.to(dev())), y, olen, ylen, C - 1)
	assert torch.isinf(nll[1]) and nll[1] > 0 and float(grad[1].abs().max()) == 0.0
	close(nll.cpu()[[0, 2]], ref.detach().float()[[0, 2]], 1e-5, 1e-4, 'nll')
	g0 = grad.cpu()[0]
	keep = torch.ones(C, dtype = torch.bool); keep[C - 2] = False
	close(g0[keep], lpr.grad.float()[0][keep], 1e-4, 2e-4, 'grad beside a forbidden class')
	close(grad.cpu()[2], lpr.grad.float()[2], 1e-4, 2e-4, 'grad')
	lp[2, :, 100] = float('nan')   # what log_softmax makes of a frame with a NaN logit
	nll, grad = ops.ctc_loss(ops.as_cl(lp.to(dev())), y, olen, ylen, C - 1)   # returns
	torch.cuda.synchronize()
	assert not torch.isfinite(nll[2]) and torch.isfinite(nll[0])


@gpu
def test_ctc_repeated_launches_are_bitwise_identical():
	"""The two waves of a CTC sweep hand their edge states over through polled LDS slots: nothing in the arithmetic depends on their
	relative timing, so every launch must reproduce the first one bit for bit (a difference is a race).  A memory-bound kernel on another
	stream perturbs the timing.  (scratch/ctc_soak.py: 18,000 launches over six shapes, 0 mismatches.)"""
	from convasr_amd import ops
	d = dev()
	for (B, T, C, S) in [(64, 753, 38, 150), (8, 900, 129, 383), (6, 2100, 38, 1000)]:
		torch.manual_seed(S)
		lp = (torch.randn(B, T, C, device = d) * 2).log_softmax(-1).contiguous().transpose(1, 2)
		y = torch.randint(0, C - 1, (B, S), device = d)
		olen = torch.randint(max(T // 2, 2 * S + 1), T + 1, (B, ), device = d)
		ylen = torch.randint(max(S // 2, 1), S + 1, (B, ), device = d)
		nll0, g0 = ops.ctc_loss(lp, y, olen, ylen, C - 1)
		nll0, g0 = nll0.clone(), g0.clone()
		assert torch.isfinite(nll0).all()
		side, junk = torch.cuda.Stream(), torch.empty(32 << 20, device = d)
		for i in range(150):
			if i % 3 == 0:
				with torch.cuda.stream(side):
					junk.add_(1.0)
			nll, g = ops.ctc_loss(lp, y, olen, ylen, C - 1)
			assert torch.equal(nll, nll0) and torch.equal(g, g0), i
		torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ optimizer

@gpu
def test_sumsq_and_sgd_step_match_torch():
	from convasr_amd import ops
	torch.manual_seed(0)
	n = 1_000_003
	p0, g0 = torch.randn(n), torch.randn(n) * 0.5
	p = torch.nn.Parameter(p0.clone())
	opt = torch.optim.SGD([p], lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	d = dev()
	pd, gd, buf = p0.to(d), g0.to(d), torch.zeros(n, device = d)
	for it in range(3):
		p.grad = g0.clone() * (it + 1)
		norm = torch.nn.utils.clip_grad_norm_([p], 100.0)
		opt.step()
		gi = gd * (it + 1)
		ss = ops.sumsq(gi)
		close(ss.sqrt().float().squeeze(), norm, 5e-5, 0, 'grad norm')
		ops.sgd_step(pd, gi, buf, n, ss, 100.0, 1e-2, 0.9, 1e-3, False, it == 0)
		close(pd, p.detach(), 1e-5, 1e-6, f'params after step {it}')
	# device-side loss gate (train.py:769-772 without the host round trip): a non-finite loss leaves parameters and momentum alone
	before, mom = pd.clone(), buf.clone()
	for bad in (float('inf'), float('nan'), -float('inf')):
		ops.sgd_step(pd, gd, buf, n, ss, 100.0, 1e-2, 0.9, 1e-3, False, False, loss_gate = torch.tensor([bad], device = d))
		assert torch.equal(pd, before) and torch.equal(buf, mom)
	ops.sgd_step(pd, gd, buf, n, ss, 100.0, 1e-2, 0.9, 1e-3, False, False, loss_gate = torch.tensor([3.5], device = d))
	assert not torch.equal(pd, before)
	# grad_scale: rank-summed gradients with 1 / world folded into the kernel == the averaged gradient fed plainly (clip included)
	pa, pb, ba, bb = p0.to(d), p0.to(d), torch.zeros(n, device = d), torch.zeros(n, device = d)
	gsum = gd * 400.0  # large enough for max_norm = 100 to clip after the 1/4 scaling
	ops.sgd_step(pa, gsum, ba, n, ops.sumsq(gsum), 100.0, 1e-2, 0.9, 1e-3, False, True, grad_scale = 0.25)
	gavg = gsum * 0.25
	ops.sgd_step(pb, gavg, bb, n, ops.sumsq(gavg), 100.0, 1e-2, 0.9, 1e-3, False, True)
	close(pa, pb, 1e-6, 1e-7, 'grad_scale')
	close(ba, bb, 1e-5, 1e-6, 'grad_scale momentum')


@gpu
def test_train_step_device_gate_skips_on_infeasible_target():
	"""An utterance whose target cannot be aligned (more labels than frames) gives an inf CTC loss: the reference skips the
	iteration; here backward runs but the optimizer kernel must leave every parameter untouched, on the device-gated path and
	on the host-gated one alike."""
	import convasr_amd as ca
	d = dev()
	for device_gate in (True, False):
		torch.manual_seed(0)
		model = ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.0], out_width_factors_large = [2, 2], residual = False, repeat = 1, check_time_dim_padded = False, temporal_mask = False).to(d).train()
		flat = ca.train.FlatParameters(model)
		model._convasr_flat = flat
		opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
		x, xlen = torch.randn(2, 64, 40, device = d), torch.ones(2, device = d)
		y = torch.randint(0, 37, (2, 1, 30), device = d)
		ok_len, bad_len = torch.tensor([[5], [4]], device = d), torch.tensor([[5], [30]], device = d)  # 30 labels in 20 output frames
		before = flat.data.clone()
		res = ca.train.train_step(model, opt, x, xlen, y, bad_len, device_gate = device_gate)
		assert bool(res['skipped']) and torch.equal(flat.data, before)
		res = ca.train.train_step(model, opt, x, xlen, y, ok_len, device_gate = device_gate)
		assert not bool(res['skipped']) and not torch.equal(flat.data, before) and torch.isfinite(flat.data).all()


# ------------------------------------------------------------------------------------------------ full-size, size-independent properties

@gpu
@pytest.mark.parametrize('layer', [(768, 768, 11, 1), (768, 896, 29, 2), (896, 1024, 1, 1), (256, 384, 11, 1)])
def test_conv_adjoint_identities_at_benchmark_size(layer):
	"""BASELINE configs[2] size (64 x 751 frames, bf16): forward, dgrad and wgrad must be three views of one bilinear form,
	<conv(x, w), dy> == <x, dgrad(dy, w)> == <w, wgrad(x, dy)>, whatever the tiling / split-K / LDS-DMA schedule does."""
	from convasr_amd import ops, _lib
	cin, cout, k, dil = layer
	torch.manual_seed(cin + cout + k)
	d = dev()
	B, T = 64, 751
	pad = dil * k // 2
	x = ops.as_cl(torch.randn(B, cin, T, device = d), torch.bfloat16)
	w = (torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5).bfloat16().float()
	fwd, dgr = ops.pack_weight(w, torch.bfloat16, None)
	Tout = ops.conv_out_len(T, k, 1, dil, pad)
	dy = ops.as_cl(torch.randn(B, cout, Tout, device = d), torch.bfloat16)
	y = ops.conv1d(x, fwd, cout, k, 1, dil, pad, out_dtype = torch.float32)
	dx = ops.conv1d(dy, dgr, cin, k, 1, dil, dil * (k - 1) - pad, out_dtype = torch.float32)
	dw = torch.empty_like(w)
	ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
	a = float((y.double() * dy.double()).sum())
	b = float((x.double() * dx.double()).sum())
	c = float((w.double() * dw.double()).sum())
	scale = float(y.double().norm() * dy.double().norm())
	assert abs(a - b) / scale < 2e-5 and abs(a - c) / scale < 2e-5, (a, b, c, scale)
	# linearity of the forward map in x at full size
	x2 = ops.as_cl(torch.randn(B, cin, T, device = d), torch.bfloat16)
	y2 = ops.conv1d(x2, fwd, cout, k, 1, dil, pad, out_dtype = torch.float32)
	xs = ops.as_cl((x.float() + x2.float()).bfloat16(), torch.bfloat16)
	ys = ops.conv1d(xs, fwd, cout, k, 1, dil, pad, out_dtype = torch.float32)
	err = (ys - (y + y2)).abs().max() / (y.abs().max() + y2.abs().max())
	assert float(err) < 2e-2  # bf16 rounding of x + x2


@gpu
def test_ctc_properties_at_benchmark_size():
	"""64 x 753 frames x 38 classes, 150 labels: posteriors sum to one per frame (gradient rows sum to zero), the gradient is
	zero beyond olen, and shifting all log-probs of an utterance by a constant c changes its NLL by exactly -olen * c."""
	from convasr_amd import ops
	torch.manual_seed(0)
	d = dev()
	B, C, T, S = 64, 38, 753, 150
	lp = torch.randn(B, C, T, device = d).log_softmax(dim = 1)
	y = torch.randint(0, C - 1, (B, S), device = d)
	olen = torch.randint(2 * S + 1, T + 1, (B, ), device = d)
	ylen = torch.randint(S // 2, S + 1, (B, ), device = d)
	nll, grad = ops.ctc_loss(ops.as_cl(lp), y, olen, ylen, C - 1)
	assert torch.isfinite(nll).all()
	rows = grad.sum(dim = 1)  # exp(lp) sums to 1 and the posterior sums to 1 for every valid frame
	assert float(rows.abs().max()) < 5e-3  # alpha + beta + nll are O(2800) in fp32 (ulp 2.4e-4): ~1e-3 relative after exp, as in ATen
	tmask = torch.arange(T, device = d)[None, :] >= olen[:, None]
	assert float((grad.abs().amax(dim = 1) * tmask).max()) == 0.0
	nll2, _ = ops.ctc_loss(ops.as_cl(lp - 0.25), y, olen, ylen, C - 1, need_grad = False)
	close(nll2 - nll, 0.25 * olen.float(), 1e-4, 1e-2, 'shift property')


@gpu
def test_loss_head_matches_the_reference_expressions():
	"""train.py:754-756 + 769: loss = (loss * ylen[:, 0]).mean() / accum, loss_cur = loss.mean(), entropy mean, the inf/NaN flag, and
	the gradient autograd would hand back to the loss vector."""
	from convasr_amd import ops
	d = dev()
	g = torch.Generator().manual_seed(3)
	for B, accum in ((64, 1), (7, 4), (300, 2)):
		lv = (torch.rand(B, generator = g) * 5 + 0.1).requires_grad_(True)
		ylen = torch.randint(1, 200, (B, 2), generator = g)
		ent = torch.rand(B, generator = g)
		loss = (lv * ylen[:, 0]).mean() / accum
		loss.backward()
		out3, gvec, skipped = ops.loss_head(lv.detach().to(d), ylen.to(d)[:, 0], ent.to(d), accum)
		close(out3, torch.stack([loss.detach(), lv.detach().mean(), ent.mean()]), 2e-6, 1e-7, 'loss head scalars')
		close(gvec, lv.grad, 1e-6, 0, 'd loss / d loss_vec')
		assert not bool(skipped)
	bad = lv.detach().clone(); bad[3] = float('inf')
	assert bool(ops.loss_head(bad.to(d), ylen.to(d)[:, 0], None, 1)[2])
	bad[3] = float('nan')
	assert bool(ops.loss_head(bad.to(d), ylen.to(d)[:, 0], None, 1)[2])
	# clip_grad_norm_'s return value straight from the reduction kernel
	gflat = torch.randn(10007, generator = g).to(d)
	norm = torch.empty(1, device = d)
	ops.sumsq(gflat, norm_out = norm, norm_scale = 0.5)
	close(norm[0], gflat.double().norm() * 0.5, 1e-6, 0, 'grad norm')
	# nll / ylen[:, 0] folded into the CTC function: value and gradient equal the two-step form
	from convasr_amd import functional as Fn
	B, C, T, S = 4, 38, 60, 9
	lp = torch.randn(B, C, T, generator = g).log_softmax(dim = 1).to(d)
	y = torch.randint(0, 37, (B, S), generator = g).to(d)
	olen, yl = torch.tensor([60, 50, 41, 33]).to(d), torch.tensor([[9], [7], [5], [3]]).to(d)
	a = ops.as_cl(lp).clone().requires_grad_(True)
	b_ = ops.as_cl(lp).clone().requires_grad_(True)
	la = Fn.ctc_loss(a, y, olen, yl[:, 0], 37, norm = yl[:, 0])
	lb = Fn.ctc_loss(b_, y, olen, yl[:, 0], 37) / yl[:, 0]
	w = torch.rand(B, generator = g).to(d)
	la.backward(w); lb.backward(w)
	close(la, lb, 1e-7, 0, 'normalised CTC loss')
	close(a.grad, b_.grad, 1e-6, 1e-9, 'normalised CTC gradient')


@gpu
@pytest.mark.parametrize('case', [(256, 64, 11, 5, 1502), (128, 64, 13, 6, 600), (256, 128, 3, 1, 258), (128, 64, 5, 0, 400), (128, 64, 4, 2, 320)])
def test_stride2_fold_forward_and_weight_gradient(case):
	"""The stride-2 fold (include/convasr_hip.h): the folded stride-1 problem over the (T / 2, 2 Cin) view gives F.conv1d(stride = 2)'s
	output and weight gradient -- against torch fp32 on the bf16-rounded operands, and against the general strided kernel; the folded
	operand and the unfolded gradient are exact rearrangements (bitwise) in both parameter layouts."""
	from convasr_amd import ops, _lib
	import torch.nn.functional as F
	Cout, Cin, K, pad, T = case
	d = dev()
	torch.manual_seed(K * 100 + pad)
	B = 3
	w = torch.randn(Cout, Cin, K, device = d) / (Cin * K) ** 0.5
	wk = torch.empty(K, Cout, Cin, device = d).permute(1, 2, 0)
	wk.copy_(w)
	Kf, Pf = ops.fold2_geometry(K, pad)
	s0 = 2 * Pf - pad
	assert Pf == (pad + 1) // 2 and Kf == (K - 1 + s0) // 2 + 1
	# the packed operand is w rearranged
	expect = torch.zeros(Kf, Cout, 2, Cin, device = d)
	for j in range(Kf):
		for p in range(2):
			k = 2 * j + p - s0
			if 0 <= k < K:
				expect[j, :, p, :] = w[:, :, k]
	for src in (w, wk):
		for dt in (torch.float32, torch.bfloat16):
			wp = ops.fold2_pack_weight(src, dt, pad)
			assert wp.shape == (Kf, ops.cout_pad(Cout), 2 * Cin)
			assert torch.equal(wp[:, :Cout], expect.view(Kf, Cout, 2 * Cin).to(dt)) and not bool(wp[:, Cout:].any())
	x = ops.as_cl(torch.randn(B, Cin, T, device = d), torch.bfloat16)
	xv = x.as_strided((B, 2 * Cin, T // 2), (T * Cin, 1, 2 * Cin))
	Tout = ops.conv_out_len(T, K, 2, 1, pad)
	wb = w.to(torch.bfloat16).float()
	ref = F.conv1d(x.float(), wb, stride = 2, padding = pad)
	assert ref.shape[2] == Tout
	y = ops.conv1d(xv, ops.fold2_pack_weight(wk, torch.bfloat16, pad), Cout, Kf, 1, 1, Pf, Tout = Tout, out_dtype = torch.float32)
	close(y, ref, 1e-3, 1e-3, 'folded forward vs torch')
	y0 = ops.conv1d(x, ops.pack_weight(w, torch.bfloat16, _lib.PACK_FWD), Cout, K, 2, 1, pad, out_dtype = torch.float32)
	close(y, y0, 1e-3, 1e-3, 'folded forward vs the strided kernel')
	# weight gradient
	dy = ops.as_cl(torch.randn(B, Cout, Tout, device = d), torch.bfloat16)
	xr = x.float().requires_grad_()
	wr = wb.clone().requires_grad_()
	F.conv1d(xr, wr, stride = 2, padding = pad).backward(dy.float())
	dwf = torch.empty(Kf, Cout, 2 * Cin, device = d)
	ops.conv1d_wgrad(xv, dy, Cout, Kf, 1, 1, Pf, dwf.permute(1, 2, 0))
	for layout in ('reference', 'kmajor'):
		dw = torch.full((Cout, Cin, K), 2.0, device = d) if layout == 'reference' else torch.full((K, Cout, Cin), 2.0, device = d).permute(1, 2, 0)
		ops.fold2_unfold_wgrad(dwf, dw, pad)
		for k in range(K):
			j, p = (k + s0) // 2, (k + s0) % 2
			assert torch.equal(dw[:, :, k], dwf[j, :, p * Cin:(p + 1) * Cin]), (layout, k)
		close(dw, wr.grad, 2e-3, 2e-3 * float(wr.grad.abs().max()), 'folded wgrad vs torch')
		before = dw.clone()
		ops.fold2_unfold_wgrad(dwf, dw, pad, accumulate = True)
		assert torch.equal(dw.contiguous(), (2 * before).contiguous())
	dw0 = torch.empty(Cout, Cin, K, device = d)
	ops.conv1d_wgrad(x, dy, Cout, K, 2, 1, pad, dw0)
	close(dw, 2 * dw0, 2e-3, 4e-3 * float(dw0.abs().max()), 'folded wgrad vs the strided kernel')


@gpu
def test_stride2_fold_in_the_model_matches_the_strided_kernels():
	"""Wav2Letter bf16 training step with the prologue folded (default) against the same step with the general strided kernels
	(Fold2.enabled = False): same loss and prologue gradient up to bf16 summation order; odd frame counts are padded by one zero frame."""
	import convasr_amd
	from convasr_amd import models, functional as Fn
	d = dev()

	def run(enabled):
		Fn.Fold2.enabled = enabled
		try:
			torch.manual_seed(3)
			Fn.manual_seed(11)
			m = models.Wav2Letter(64, [38], dropout = 0.0, compute_dtype = torch.bfloat16, check_time_dim_padded = False).to(d).train()
			x = torch.randn(2, 64, 301, device = d)
			xlen = torch.tensor([1.0, 0.7], device = d)
			y = torch.randint(0, 37, (2, 1, 20), device = d)
			ylen = torch.tensor([[20], [12]], device = d)
			out = m(x, xlen, y = y, ylen = ylen)
			out['loss'].sum().backward()
			w0 = m.backbone[0].conv[0][-1].weight
			return out['loss'].detach().clone(), w0.grad.detach().clone(), out['log_probs'][0].detach().clone()
		finally:
			Fn.Fold2.enabled = True

	l1, g1, lp1 = run(True)
	l0, g0, lp0 = run(False)
	assert lp1.shape == lp0.shape
	close(l1, l0, 2e-2, 1e-2, 'loss, folded vs strided prologue')
	cos = float((g1.double().flatten() @ g0.double().flatten()) / (g1.double().norm() * g0.double().norm()))
	assert cos > 0.999, cos


@gpu
def test_instnorm_time_padding_and_folded_eval_forward():
	"""ops.instnorm(pad_time_to = 2): the first T frames equal the unpadded result bit for bit, the appended frame is zero; the fused eval
	forward (BN folded into the conv epilogue) of a model whose prologue runs through the stride-2 fold matches the strided kernels."""
	from convasr_amd import ops, models, functional as Fn
	d = dev()
	torch.manual_seed(5)
	x = torch.randn(3, 64, 301, device = d)
	xlen = torch.tensor([1.0, 0.6, 0.83], device = d)
	for dt in (torch.float32, torch.bfloat16):
		a = ops.instnorm(x, xlen, 1e-5, out_dtype = dt)
		b = ops.instnorm(x, xlen, 1e-5, out_dtype = dt, pad_time_to = 2)
		assert b.shape == (3, 64, 302) and ops.is_cl(b) and torch.equal(b[:, :, :301], a) and not bool(b[:, :, 301].any())
		assert ops.instnorm(x[:, :, :300], xlen, 1e-5, out_dtype = dt, pad_time_to = 2).shape == (3, 64, 300)

	def run(enabled):
		Fn.Fold2.enabled = enabled
		try:
			torch.manual_seed(3)
			m = models.Wav2Letter(64, [38], dropout = 0.0, compute_dtype = torch.bfloat16, check_time_dim_padded = False).to(d)
			m.train()
			with torch.no_grad():
				m(x, xlen)  # running statistics
			m.eval()
			m.fuse_conv_bn_eval()
			with torch.no_grad():
				return m(x, xlen)['log_probs'][0].float()
		finally:
			Fn.Fold2.enabled = True

	l1, l0 = run(True), run(False)
	assert l1.shape == l0.shape
	err = float((l1 - l0).norm() / l0.norm())
	assert err < 2e-2, err  # two bf16 pipelines that differ in one layer's summation order


@gpu
@pytest.mark.parametrize('act', ['hardtanh', 'relu'])
def test_stored_gradient_gates_equal_the_rederived_ones(act):
	"""The one-bit gates convasr_bn_act_fwd stores (include/convasr_hip.h): both backward consumers -- the streaming apply pass and the
	fused dgrad epilogue -- produce bit-identical results from the bits and from re-deriving act' / dropout / frame mask."""
	from convasr_amd import ops, _lib
	d = dev()
	torch.manual_seed(7)
	B, C, T, Cout, K = 3, 256, 300, 128, 5
	a = (_lib.ACT_HARDTANH, 0.0, 20.0) if act == 'hardtanh' else (_lib.ACT_RELU, 0.0, 0.0)
	y = ops.as_cl(torch.randn(B, C, T, device = d) * 8, torch.bfloat16)
	scale, shift = torch.rand(C, device = d) + 0.5, torch.randn(C, device = d) * 4 + 4
	mean, invstd = torch.randn(C, device = d), torch.rand(C, device = d) + 0.5
	xlen = torch.tensor([1.0, 0.55, 0.8], device = d)
	for p_drop in (0.0, 0.3):
		gate = torch.zeros(B * T * C // 8, dtype = torch.uint8, device = d)
		z = ops.bn_act(y, scale, shift, a, xlen = xlen, dropout_p = p_drop, seed = 5, offset = 11, gate = gate)
		z0 = ops.bn_act(y, scale, shift, a, xlen = xlen, dropout_p = p_drop, seed = 5, offset = 11)
		assert torch.equal(z, z0)
		bits = ((gate.view(B, T, C // 8, 1) >> torch.arange(8, device = d, dtype = torch.uint8)) & 1).reshape(B, T, C).permute(0, 2, 1).bool()
		nz = z.float() != 0
		assert bool((bits <= nz).all()) and float(bits.float().mean()) > 0.05  # a passing gradient implies a non-zero output (the converse fails at the upper clamp)
		frames = (xlen * T).ceil().long()
		for b_ in range(B):
			assert not bool(bits[b_, :, int(frames[b_]):].any())
		# streaming apply pass
		dz = ops.as_cl(torch.randn(B, C, T, device = d), torch.bfloat16)
		coef = torch.randn(3 * C, device = d)
		dy0 = ops.bn_act_bwd_apply(dz, y, coef, True, scale, shift, a, xlen = xlen, dropout_p = p_drop, seed = 5, offset = 11)
		dy1 = ops.bn_act_bwd_apply(dz, y, coef, True, scale, shift, a, xlen = xlen, dropout_p = p_drop, seed = 5, offset = 11, gate = gate)
		assert torch.equal(dy0, dy1)
		# fused dgrad epilogue: dx identical, partial sums identical
		dyc = ops.as_cl(torch.randn(B, Cout, T, device = d), torch.bfloat16)
		w = torch.randn(Cout, C, K, device = d) / (C * K) ** 0.5
		_, wd = ops.pack_weight(w, torch.bfloat16, None)
		s0, s1 = ops.ConvStats(C, B, T, d), ops.ConvStats(C, B, T, d)
		dx0 = ops.conv1d_dgrad_bn_reduce(dyc, wd, C, K, 1, K - 1 - K // 2, y, scale, shift, mean, invstd, a, p_drop, 5, 11, xlen, s0)
		dx1 = ops.conv1d_dgrad_bn_reduce(dyc, wd, C, K, 1, K - 1 - K // 2, y, scale, shift, mean, invstd, a, p_drop, 5, 11, xlen, s1, gate = gate)
		assert dx0 is not None and torch.equal(dx0, dx1) and s0.rows == s1.rows
		# the two forms of the fused epilogue sum the same g (vector ALU, per-thread partial sums / matrix pipe, one fp32 chain per tile)
		close(s1.totals()[:C], s0.totals()[:C], 1e-5, 1e-3, 'fused epilogue, sum g: matrix-pipe form vs vector-ALU form')
		close(s1.totals()[C:], s0.totals()[C:], 1e-5, 3e-3, 'fused epilogue, sum g xhat: matrix-pipe form vs vector-ALU form')
		# separate reduce pass (the unfused form): coefficients and parameter gradients from the bits against the re-derived ones
		# (same g per element; the gated kernel sums g y and centres once per block, the other sums g xhat per element: fp32 rounding apart)
		gamma = torch.rand(C, device = d) + 0.5
		outs = []
		for gt in (None, gate):
			coef_r, dgm, dbt = torch.empty(3 * C, device = d), torch.empty(C, device = d), torch.empty(C, device = d)
			ops.bn_act_bwd_reduce(dz, y, scale, shift, mean, invstd, a, xlen = xlen, dropout_p = p_drop, seed = 5, offset = 11, write_g = False, gamma = gamma, coef = coef_r, dgamma = dgm, dbeta = dbt, gate = gt)
			outs.append((coef_r, dgm, dbt))
		for u, v, what in zip(outs[0], outs[1], ('coef', 'dgamma', 'dbeta')):
			close(v, u, 1e-4, 1e-3 if what != 'coef' else 1e-5, 'gated reduce ' + what)


@gpu
def test_training_step_is_the_same_with_and_without_stored_gates():
	"""Stored gates against re-derived ones over a whole bf16 training step.  The forward pass is bit-identical; in the backward pass the
	fused dgrad epilogue sums g and g y on the matrix pipe when it has the gates (conv_v2s.hip, form 2) and on the vector ALU when it
	re-derives them: the same per-element g (bitwise, test above), fp32 sums in a different order, so the batch-norm coefficients
	agree to ~1e-6 and a few bf16 roundings of dy flip downstream: gradients agree to 2 % of their largest entry, cosine > 0.999."""
	from convasr_amd import models, functional as Fn
	d = dev()

	def run(enabled):
		Fn.GATE_BITS = enabled
		try:
			torch.manual_seed(3)
			Fn.manual_seed(11)
			m = models.Wav2Letter(64, [38], dropout = 0.2, compute_dtype = torch.bfloat16, check_time_dim_padded = False).to(d).train()
			x = torch.randn(2, 64, 302, device = d)
			xlen = torch.tensor([1.0, 0.7], device = d)
			y = torch.randint(0, 37, (2, 1, 20), device = d)
			ylen = torch.tensor([[20], [12]], device = d)
			out = m(x, xlen, y = y, ylen = ylen)
			out['loss'].sum().backward()
			return out['loss'].detach().clone(), [p.grad.detach().clone() for p in m.parameters()]
		finally:
			Fn.GATE_BITS = True

	l1, g1 = run(True)
	l0, g0 = run(False)
	assert torch.equal(l1, l0)
	for a, b in zip(g1, g0):
		a, b = a.double().flatten(), b.double().flatten()
		assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-12 and (float(b.norm()) == 0 or float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.999)


@gpu
def test_stride2_fold_adjoint_identity_at_the_bench_size():
	"""The folded prologue at BASELINE's full size (64 utterances x 1502 frames x 64 mel channels -> 256 channels, K = 11, stride 2):
	<conv(x; w), dy> = <w, wgrad(x, dy)> -- the forward and the weight gradient are adjoint in w -- evaluated in float64 on the kernels'
	bf16 / fp32 outputs (size-independent property: no oracle run needed at this size)."""
	from convasr_amd import ops
	d = dev()
	torch.manual_seed(21)
	B, Cin, Cout, K, pad, T = 64, 64, 256, 11, 5, 1502
	x = ops.as_cl(torch.randn(B, Cin, T, device = d), torch.bfloat16)
	w = (torch.randn(Cout, Cin, K, device = d) / (Cin * K) ** 0.5).to(torch.bfloat16).float()  # exactly representable: the packed operand is w itself
	Kf, Pf = ops.fold2_geometry(K, pad)
	Tout = ops.conv_out_len(T, K, 2, 1, pad)
	xv = x.as_strided((B, 2 * Cin, T // 2), (T * Cin, 1, 2 * Cin))
	y = ops.conv1d(xv, ops.fold2_pack_weight(w, torch.bfloat16, pad), Cout, Kf, 1, 1, Pf, Tout = Tout, out_dtype = torch.float32)
	assert y.shape == (B, Cout, Tout)
	dy = ops.as_cl(torch.randn(B, Cout, Tout, device = d), torch.bfloat16)
	dwf = torch.empty(Kf, Cout, 2 * Cin, device = d)
	ops.conv1d_wgrad(xv, dy, Cout, Kf, 1, 1, Pf, dwf.permute(1, 2, 0))
	dw = torch.empty(Cout, Cin, K, device = d)
	ops.fold2_unfold_wgrad(dwf, dw, pad)
	lhs = float((y.double() * dy.double()).sum())
	rhs = float((w.double() * dw.double()).sum())
	scale = float(y.double().norm() * dy.double().norm())
	assert abs(lhs - rhs) <= 1e-5 * scale, (lhs, rhs, scale)
	# linearity in x of the folded forward
	x2 = ops.as_cl(torch.randn(B, Cin, T, device = d), torch.bfloat16)
	xs = ops.as_cl((x.float() + x2.float()).to(torch.bfloat16).float(), torch.float32)  # the sum as the kernel will see it (bf16)
	wp = ops.fold2_pack_weight(w, torch.bfloat16, pad)
	f = lambda t: ops.conv1d(ops.as_cl(t, torch.bfloat16).as_strided((B, 2 * Cin, T // 2), (T * Cin, 1, 2 * Cin)), wp, Cout, Kf, 1, 1, Pf, Tout = Tout, out_dtype = torch.float32)
	ya, yb_, ys = f(x), f(x2), f(xs.to(torch.bfloat16))
	exact = (ops.as_cl(xs, torch.float32) - x.float() - x2.float()).abs().max()  # rounding of the bf16 sum
	assert float((ys - ya - yb_).abs().max()) <= 64 * K * float(exact) * float(w.abs().max()) + 1e-3


@gpu
@pytest.mark.parametrize('dtype', ['bf16', 'f16'])
def test_fused_dgrad_epilogue_matrix_pipe_sums_at_the_bench_size(dtype):
	"""The fused BN-backward epilogue with stored gates (conv_v2s.hip form 2: sums on the matrix pipe) at a size that runs BOTH tile
	widths in one launch (64 x 751 frames, 256 channels: 384 tiles = 256 full + 128 cut into halves): dx bit-identical to the plain
	dgrad launch, sum g and sum g xhat equal to the stand-alone gated reduce pass (fp32 sums in a different order) -- bf16 and fp16."""
	from convasr_amd import ops, _lib
	d = dev()
	torch.manual_seed(5)
	dt = HALF[dtype][0]
	B, T, C, Cout, K = 64, 751, 256, 256, 5
	dy = ops.as_cl(torch.randn(B, Cout, T, device = d), dt)
	w = torch.randn(Cout, C, K, device = d) / (C * K) ** 0.5
	_, wd = ops.pack_weight(w, dt, None)
	y = ops.as_cl(torch.randn(B, C, T, device = d) * 4 + 2, dt)
	scale, shift = torch.rand(C, device = d) + 0.5, torch.randn(C, device = d)
	mean, invstd = torch.randn(C, device = d), torch.rand(C, device = d) + 0.5
	xlen = torch.linspace(0.5, 1, B, device = d)
	act = (_lib.ACT_HARDTANH, 0.0, 20.0)
	gate = torch.zeros(B * T * C // 8, dtype = torch.uint8, device = d)
	ops.bn_act(y, scale, shift, act, xlen = xlen, dropout_p = 0.2, seed = 3, offset = 5, gate = gate)
	pad = K - 1 - K // 2
	sums = ops.ConvStats(C, B, T, d)
	dx = ops.conv1d_dgrad_bn_reduce(dy, wd, C, K, 1, pad, y, scale, shift, mean, invstd, act, 0.2, 3, 5, xlen, sums, gate = gate)
	assert dx is not None and torch.equal(dx, ops.conv1d(dy, wd, C, K, 1, 1, pad))
	ref = torch.zeros(2 * C, dtype = torch.float64, device = d)
	ops.bn_act_bwd_reduce(dx, y, scale, shift, mean, invstd, act, xlen = xlen, dropout_p = 0.2, seed = 3, offset = 5, sums = ref, write_g = False, gate = gate)
	got = sums.totals()
	close(got[:C], ref[:C], 1e-5, 2e-2, 'sum g')  # sums of ~48 K terms of magnitude ~1: fp32 chains in different orders
	close(got[C:], ref[C:], 1e-5, 1e-1, 'sum g xhat')
	assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16, torch.float16])
def test_colsum_is_the_bias_gradient_of_a_conv(dt):
	"""convasr_colsum (the split-operand head's bias gradient, models.py:26): out[c] (+)= sum over (b, t) of a channels-last tensor, fp32 sums in a fixed
	order -- against a float64 sum, bitwise repeatable, accumulate adds."""
	from convasr_amd import ops
	d = torch.device('cuda:0')
	torch.manual_seed(0)
	y = ops.as_cl((torch.randn(5, 38, 777) * 3).to(d), dt)
	ref = y.double().sum(dim = (0, 2))
	out = torch.empty(38, device = d)
	ops.colsum(y, out)
	again = torch.empty(38, device = d)
	ops.colsum(y, again)
	assert torch.equal(out, again)
	assert float((out.double() - ref).abs().max()) <= 2e-6 * float(y.double().abs().sum(dim = (0, 2)).max())
	ops.colsum(y, out, accumulate = True)
	assert torch.allclose(out.double(), 2 * ref, rtol = 1e-5, atol = 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('half', [torch.float16, torch.bfloat16])
def test_cast_scale_packs_and_unpacks_a_gradient_bucket(half):
	"""convasr_cast_scale, the two ends of the 16-bit gradient exchange (models.py:744-765 under apex O2): fp32 -> 16-bit x scale with
	round-to-nearest-even (inf beyond fp16's range: what the loss scaler's overflow check keys on), and back exactly."""
	from convasr_amd import _lib
	d = torch.device('cuda:0')
	torch.manual_seed(1)
	n = 8 * 1000
	g = (torch.randn(n) * torch.logspace(-6, 3, n)).to(d)
	g[5] = 1e6
	buf = torch.empty(n, dtype = half, device = d)
	back = torch.empty(n, device = d)
	s = _lib.stream_ptr()
	_lib.call('convasr_cast_scale', _lib.ptr(g), _lib.F32, _lib.ptr(buf), _lib.dtype_code(half), n, 0.125, s)
	_lib.call('convasr_cast_scale', _lib.ptr(buf), _lib.dtype_code(half), _lib.ptr(back), _lib.F32, n, 1.0, s)
	want = (g * 0.125).to(half)
	assert torch.equal(buf, want) and torch.equal(back, want.float())
	assert bool(torch.isinf(buf[5])) == (half == torch.float16)
	with pytest.raises(_lib.ConvasrHipError):
		_lib.call('convasr_cast_scale', _lib.ptr(g), _lib.F32, _lib.ptr(buf), _lib.dtype_code(half), n - 3, 1.0, s)  # n % 8 != 0
