"""Tensor-level wrappers over the C ABI (no autograd here; see functional.py).

Activations travel as torch tensors of LOGICAL shape (B, C, T) -- the reference's convention -- whose memory is
channels-last: strides (T*C, 1, C).  `empty_cl` allocates one, `as_cl` converts anything else with the HIP layout kernel.
PyTorch is used for allocation, streams and dtype bookkeeping only; every arithmetic op is a kernel of libconvasr_hip.so.
"""
import math
import os

import ctypes

import torch

from . import _lib
from ._lib import call, ptr, stream_ptr, dtype_code, require_cuda

HALF_DTYPES = (torch.bfloat16, torch.float16)  # the two 16-bit storage types of the MFMA path (fp32 accumulation, fp32 master weights)

ACT_CODES = dict(none = _lib.ACT_NONE, relu = _lib.ACT_RELU, hardtanh = _lib.ACT_HARDTANH, leaky_relu = _lib.ACT_LEAKY_RELU)


def act_args(nonlinearity):
	"""('relu',) / ('hardtanh', lo, hi) / ('leaky_relu', slope) -> (code, lo, hi) (reference: models.py:368)."""
	if nonlinearity is None:
		return _lib.ACT_NONE, 0.0, 0.0
	name = nonlinearity[0]
	if name == 'relu':
		return _lib.ACT_RELU, 0.0, 0.0
	if name == 'hardtanh':
		return _lib.ACT_HARDTANH, float(nonlinearity[1]), float(nonlinearity[2])
	if name == 'leaky_relu':
		return _lib.ACT_LEAKY_RELU, float(nonlinearity[1]) if len(nonlinearity) > 1 else 0.01, 0.0
	raise ValueError(f'unsupported nonlinearity {nonlinearity}')


def empty_cl(B, C, T, dtype, device):
	return torch.empty(B, T, C, dtype = dtype, device = device).permute(0, 2, 1)


def zeros_cl(B, C, T, dtype, device):
	return torch.zeros(B, T, C, dtype = dtype, device = device).permute(0, 2, 1)


def is_cl(x):
	B, C, T = x.shape
	return x.stride(1) == 1 and x.stride(2) == C and (x.stride(0) == T * C or B == 1)


def convert(x, dtype, channels_last):
	"""(B, C, T) tensor of any strides -> new tensor of `dtype`, channels-last or torch-contiguous."""
	require_cuda(x)
	B, C, T = x.shape
	out = empty_cl(B, C, T, dtype, x.device) if channels_last else torch.empty(B, C, T, dtype = dtype, device = x.device)
	call('convasr_convert_layout', ptr(x), dtype_code(x.dtype), x.stride(0), x.stride(1), x.stride(2), ptr(out), dtype_code(dtype), out.stride(0), out.stride(1), out.stride(2), B, C, T, stream_ptr())
	return out


def as_cl(x, dtype = None):
	if x.__dict__.get('_convasr_planes_only') is not None:
		raise _lib.ConvasrHipError('this tensor is the placeholder of a layer output that exists as split-operand planes only (functional.ConvBnActFunction, planes_out): its values were never written -- only the split conv the network wired behind the layer may consume it (set CONVASR_NO_PLANES_OUT=1 to get real fp32 outputs)')
	dtype = dtype or x.dtype
	if x.dtype == dtype and is_cl(x):
		return x
	return convert(x, dtype, True)


def xlen_f32(xlen, device):
	if xlen is None:
		return None
	return xlen.to(device = device, dtype = torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ frontend / instance norm

def logmel(signal, xlen, window, mel_weight, mel_bias, nfft, hop, preemphasis = 0.97, normalize = True, denom_multiplier = 1.0):
	"""LogFilterBankFrontend.forward (models.py:565-597) -> channels-last (B, nmel, F) fp32.
	denom_multiplier: models.py:570's debug_short_long_records_normalize_signal_multiplier, x / ((absmax + 1e-5) * m)."""
	require_cuda(signal, window, mel_weight, mel_bias)
	assert signal.ndim == 2
	if signal.dtype not in (torch.float32, torch.int16):
		signal = signal.to(torch.float32) if signal.is_floating_point() else signal.to(torch.int16) if signal.dtype in (torch.int8, torch.uint8) else signal.to(torch.float32)
	signal = signal.contiguous()
	B, T = signal.shape
	nmel = mel_weight.shape[0]
	F = 1 + T // hop
	absmax = None
	s = stream_ptr()
	if normalize:
		absmax = torch.empty(B, dtype = torch.float32, device = signal.device)
		call('convasr_signal_absmax', ptr(signal), dtype_code(signal.dtype), B, T, ptr(absmax), s)
		if denom_multiplier != 1.0:
			# the kernel divides by (absmax + 1e-5): hand it the peak that makes that (absmax + 1e-5) * m.  A debugging knob of the reference, off by
			# default: two B-element torch launches, not on the path of any BASELINE config
			absmax.mul_(float(denom_multiplier)).add_(1e-5 * (float(denom_multiplier) - 1.0))
	out = empty_cl(B, nmel, F, torch.float32, signal.device)
	xl = xlen_f32(xlen, signal.device)
	_lib.timed('hbm:logmel_kernel', 0.0, lambda: call('convasr_logmel_fwd', ptr(signal), dtype_code(signal.dtype), ptr(absmax), ptr(xl), ptr(window), window.shape[0], ptr(mel_weight), ptr(mel_bias), ptr(out), B, T, nfft, hop, nmel, float(preemphasis), s), nbytes = float(B * T * signal.element_size() + B * F * nmel * 4))
	return out


def instnorm(x, xlen, eps, out_dtype = None, channels_last = True, pad_time_to = 1):
	"""MaskedInstanceNorm1d.forward (models.py:694-719); xlen None = legacy unmasked branch.
	pad_time_to > 1: the result has its time axis rounded up to a multiple of it, the extra frames zero (a strided conv that follows
	sees the same zeros its own padding would have supplied; see functional.fold2)."""
	require_cuda(x)
	B, C, T = x.shape
	out_dtype = out_dtype or x.dtype
	Tp = -(-T // pad_time_to) * pad_time_to
	assert Tp == T or channels_last
	out = empty_cl(B, C, Tp, out_dtype, x.device) if channels_last else torch.empty(B, C, T, dtype = out_dtype, device = x.device)
	xl = xlen_f32(xlen, x.device)
	call('convasr_instnorm_fwd', ptr(x), dtype_code(x.dtype), x.stride(0), x.stride(1), x.stride(2), ptr(out), dtype_code(out_dtype), out.stride(0), out.stride(1), out.stride(2), ptr(xl), B, C, T, Tp, float(eps), stream_ptr())  # (the kernel writes the padding frames as zeros)
	return out


def instnorm_running(x, running_mean, running_var, num_batches_tracked, momentum, training, eps, out_dtype = None, pad_time_to = 1):
	"""nn.InstanceNorm1d(track_running_stats = True).forward (models.py:711): instance statistics + running-statistics update in training mode,
	the running statistics in eval mode; channels-last result like instnorm()."""
	require_cuda(x, running_mean, running_var)
	B, C, T = x.shape
	out_dtype = out_dtype or x.dtype
	Tp = -(-T // pad_time_to) * pad_time_to
	out = empty_cl(B, C, Tp, out_dtype, x.device)
	ws = torch.empty(B, C, 2, dtype = torch.float32, device = x.device) if training else None
	call('convasr_instnorm_running_fwd', ptr(x), dtype_code(x.dtype), x.stride(0), x.stride(1), x.stride(2), ptr(out), dtype_code(out_dtype), out.stride(0), out.stride(1), out.stride(2), B, C, T, Tp, float(eps),
		ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), float(momentum), int(bool(training)), ptr(ws), stream_ptr())
	return out


def output_lengths(xlen, B, T, device):
	"""compute_output_lengths (models.py:611-614) in one launch: int64 (B,) = ceil(xlen * T) evaluated in fp32 (xlen None: T)."""
	out = torch.empty(B, dtype = torch.int64, device = device)
	call('convasr_output_lengths', ptr(xlen_f32(xlen, device)), B, T, ptr(out), stream_ptr())
	return out


# ------------------------------------------------------------------------------------------------ conv

def cout_pad(c):
	return _lib.load().convasr_conv_cout_pad(c)


def weight_layout(w):
	"""W_KMAJOR for a (Cout, Cin, K) view of tap-major memory [K][Cout][Cin] (the training arena's layout), W_REFERENCE for a
	torch-contiguous tensor, None for anything else."""
	Cout, Cin, K = w.shape
	if w.is_contiguous():
		return _lib.W_REFERENCE  # (K == 1 is both at once: the reference layout's kernels are as good)
	if w.stride(1) == 1 and w.stride(0) == Cin and (w.stride(2) == Cout * Cin or K == 1):
		return _lib.W_KMAJOR
	return None


def pack_weight(w, dtype, mode = None, out = None, fwd_is_current = False):
	"""(Cout, Cin, K) fp32 parameter -> packed [K][rows_pad][cols] tensor(s) for the MFMA kernels.
	mode PACK_FWD returns the forward layout, PACK_DGRAD the dgrad layout, None both (fwd, dgrad).  `out` = (fwd, dgrad)
	buffers from an earlier call are refreshed in place (stable addresses, no re-allocation per optimizer step).
	fwd_is_current: out[0] already holds the parameter's packed forward copy (the optimizer's bf16 mirror): only dgrad is rebuilt."""
	require_cuda(w)
	w = w.detach()
	layout = weight_layout(w) if w.dtype == torch.float32 else None
	if layout is None:
		w, layout = w.float().contiguous(), _lib.W_REFERENCE
	Cout, Cin, K = w.shape
	fwd, dgr = out if out is not None else (None, None)
	if fwd is None:
		assert not fwd_is_current
		fwd = torch.zeros(K, cout_pad(Cout), Cin, dtype = dtype, device = w.device)
	if dgr is None and mode in (None, _lib.PACK_DGRAD):
		dgr = torch.zeros(K, cout_pad(Cin), Cout, dtype = dtype, device = w.device)
	call('convasr_pack_conv_weight', None if fwd_is_current else ptr(w), ptr(fwd), ptr(dgr) if mode in (None, _lib.PACK_DGRAD) else None, dtype_code(dtype), Cout, Cin, K, layout, stream_ptr())
	return (fwd, dgr) if mode is None else (fwd if mode == _lib.PACK_FWD else dgr)


PEAK_BF16_FLOPS, PEAK_HBM_BYTES = 2.5e15, 8.0e12  # MI355X_MICROARCH.md: dense bf16 MFMA, HBM3E


def memory_bound(flops, nbytes):
	"""Which roofline bounds a launch (for the bench's per-kernel timer only): its algorithmic bytes at 8 TB/s against its FLOPs at 2.5 PF."""
	return nbytes / PEAK_HBM_BYTES > flops / PEAK_BF16_FLOPS


def conv_out_len(Tin, K, stride, dil, pad):
	return (Tin + 2 * pad - dil * (K - 1) - 1) // stride + 1


_max_rows_cache = {}


def _conv_stats_max_rows(B, Tout):
	r = _max_rows_cache.get((B, Tout))
	if r is None:
		r = _max_rows_cache[(B, Tout)] = _lib.load().convasr_conv_stats_max_rows(B, Tout)
	return r


class ConvStats:
	"""Per-m-tile partial sums of a conv epilogue: buf is a flat fp64 tensor with room for max_rows x 2 x C, rows is how many the
	last launch wrote.  No atomics anywhere: the finalize kernels add the rows in a fixed order."""

	def __init__(self, C, B, Tout, device):
		self.C = C
		self.max_rows = _conv_stats_max_rows(B, Tout)
		self.buf = torch.empty(self.max_rows * 2 * C, dtype = torch.float64, device = device)
		self.rows = 0

	def fits(self, C, B, Tout, device):
		return self.C == C and self.buf.device == device and self.max_rows >= _conv_stats_max_rows(B, Tout)

	def totals(self):
		out = torch.empty(2 * self.C, dtype = torch.float64, device = self.buf.device)
		call('convasr_reduce_rows', ptr(self.buf), self.rows, 2 * self.C, ptr(out), stream_ptr())
		return out


def conv1d(x, wp, Cout, K, stride = 1, dil = 1, pad = 0, out_dtype = None, bias = None, stats = None, scale = None, shift = None, act = (_lib.ACT_NONE, 0.0, 0.0), xlen = None, Tout = None, work = None, family = None, splitk = False):
	"""x: channels-last (B, Cin, Tin); wp: packed weights.  Returns channels-last (B, Cout, Tout).
	splitk (the inference path's launches): a launch of a few tiles -- one online request is 4-16 workgroups per layer on 256 CUs -- is cut over the
	input channels (convasr_conv1d_fwd_splitk: fp32 partial tiles, added in split order by a second kernel that also runs the epilogue).
	Tout (optional): compute only the first Tout frames of the output; work: FLOPs to book for the bench's timer (the stride-2 fold).
	stats: None, a ConvStats (the production path: partial rows, consumed by bn_finalize), or a (2 Cout,) fp64 tensor that
	receives the totals (sum, sum of squares) -- a convenience for tests and tools, one extra tiny launch."""
	B, Cin, Tin = x.shape
	assert is_cl(x), 'conv1d expects a channels-last activation'
	Tout = conv_out_len(Tin, K, stride, dil, pad) if Tout is None else Tout
	out_dtype = out_dtype or x.dtype
	y = empty_cl(B, Cout, Tout, out_dtype, x.device)
	part = stats if isinstance(stats, ConvStats) or stats is None else ConvStats(Cout, B, Tout, x.device)
	rows = ctypes.c_int(0)
	launch = lambda: call('convasr_conv1d_fwd', ptr(x), ptr(wp), ptr(y), dtype_code(x.dtype), dtype_code(out_dtype), B, Cin, Cout, Tin, Tout, K, stride, dil, pad, ptr(bias), None if part is None else ptr(part.buf), ptr(scale), ptr(shift), act[0], act[1], act[2], ptr(xlen), ctypes.byref(rows) if part is not None else None, stream_ptr())
	if splitk and SPLITK and stats is None and stride == 1 and (x.dtype in HALF_DTYPES or (x.dtype == torch.float32 and out_dtype == torch.float32)) and Cout % 8 == 0:
		skey = (x.dtype, B, Cin, Cout, Tout, K)
		plan = _splitk_plans.get(skey)
		if plan is None:
			nb = ctypes.c_int64(0)
			plan = _splitk_plans[skey] = (_lib.load().convasr_conv1d_fwd_splitk_plan(dtype_code(x.dtype), B, Cin, Cout, Tout, K, ctypes.byref(nb)), nb.value)
		if plan[0] >= 2:
			ws = workspace(plan[1], x.device, 'splitk')
			launch = lambda: call('convasr_conv1d_fwd_splitk', ptr(x), ptr(wp), ptr(y), dtype_code(x.dtype), dtype_code(out_dtype), B, Cin, Cout, Tin, Tout, K, dil, pad, ptr(bias), ptr(scale), ptr(shift), act[0], act[1], act[2], ptr(xlen), plan[0], ptr(ws), stream_ptr())
	if _lib.timer is None:  # (the labels below are for the bench's per-kernel timer only: not on the path of an ordinary step)
		launch()
		if part is not None:
			part.rows = rows.value
			if part is not stats:
				stats.copy_(part.totals())
		return y
	# which kernel the C side picks (conv.hip: convasr_conv1d_fwd -> convasr_conv1d_v2_try), for the bench's per-kernel timer only
	label, family = family, 'conv1d_igemm_v2s_kernel<bf16>' if (x.dtype in HALF_DTYPES and stride == 1 and Cin % 64 == 0) else 'conv1d_igemm (other variants)'  # (the family name is a label: fp16 launches of the same kernel are booked under it too)
	es, osz = x.element_size(), (2 if out_dtype in HALF_DTYPES else 4)
	flops, nbytes_ = 2.0 * B * Tout * Cout * Cin * K if work is None else work, float(B * Tin * Cin * es + K * Cout * Cin * es + B * Tout * Cout * osz)
	symbol = 'v2s16' if family.startswith('conv1d_igemm_v2s') and out_dtype == x.dtype else None
	if family.startswith('conv1d_igemm_v2s') and K == 1 and pad == 0 and Cout % 128 == 0 and out_dtype == x.dtype and scale is None and act[0] == _lib.ACT_NONE and xlen is None:
		family, symbol = 'hbm:conv1x1_kernel (one-tap training launches)', None  # conv1x1.hip takes these (csrc/conv.hip: conv1d_run); bound by bytes moved, booked under the HBM roofline
	elif family.startswith('conv1d_igemm_v2s') and memory_bound(flops, nbytes_):
		family = 'hbm:conv1d_igemm_v2s_kernel (memory-bound launches: the 38-class decoder)'
	_lib.timed(label or family, flops, launch, nbytes = nbytes_, symbol = symbol)
	if part is not None:
		part.rows = rows.value
		if part is not stats:
			stats.copy_(part.totals())
	return y


SPLITK = os.environ.get('CONVASR_NO_SPLITK') != '1'  # A/B hook: small-batch inference launches cut over the input channels
_splitk_plans = {}


def _int_array(vals):
	return (ctypes.c_int * max(len(vals), 1))(*[int(v) for v in vals])


def conv1x1_grouped(xs, wps, couts, biases = None, stats = None, outs = None, accumulate = None):
	"""n one-tap convs over the same frames in one dispatch (include/convasr_hip.h, "grouped one-tap launches").  xs: channels-last
	(B, Cin_i, T) tensors of one 16-bit dtype; wps: packed weights; stats: list of ConvStats / None; outs: list of preallocated
	(B, Cout_i, T) outputs / None; accumulate: list of bools (add into outs[i]).  Returns the outputs, or None (nothing launched) when a
	problem is outside the kernel's envelope."""
	n = len(xs)
	B, _, T = xs[0].shape
	dt = xs[0].dtype
	if n == 0 or n > 12 or dt not in HALF_DTYPES or any(x.dtype != dt or x.shape[0] != B or x.shape[2] != T or not is_cl(x) or x.shape[1] % 64 for x in xs) or any(c % 128 for c in couts):
		return None
	if any(B * T * max(x.shape[1], c) * 2 >= 2 ** 31 or c * x.shape[1] * 2 >= 2 ** 31 for x, c in zip(xs, couts)):  # the kernel's 32-bit buffer offsets (csrc/conv1x1.hip): the caller launches such branches one by one
		return None
	ys = [empty_cl(B, couts[i], T, dt, xs[0].device) if outs is None or outs[i] is None else outs[i] for i in range(n)]
	rows = ctypes.c_int(0)
	cins = [x.shape[1] for x in xs]
	launch = lambda: call('convasr_conv1x1_grouped', n, _ptr_array(xs), _ptr_array(wps), _ptr_array(ys), _ptr_array(biases) if biases else None, _ptr_array([None if s_ is None else s_.buf for s_ in stats]) if stats else None,
		_int_array(cins), _int_array(couts), _int_array(accumulate) if accumulate else None, dtype_code(dt), B, T, ctypes.byref(rows), stream_ptr())
	_lib.timed('hbm:conv1x1_kernel (one-tap training launches)', 2.0 * B * T * sum(ci * co for ci, co in zip(cins, couts)), launch, nbytes = float(2 * B * T * sum(ci + co * (2 if accumulate and accumulate[i] else 1) for i, (ci, co) in enumerate(zip(cins, couts))) + 2 * sum(ci * co for ci, co in zip(cins, couts))))
	if stats:
		for s_ in stats:
			if s_ is not None:
				s_.rows = rows.value
	return ys


_wgrad_group_ws = {}


def wgrad1x1_grouped(xs, dys, dws, zeros = None, accumulate = None):
	"""dws[i] (Cout_i, Cin_i, 1) fp32 (+)= the one-tap weight gradient of (xs[i], dys[i]); zeros[i] (optional): a (Cout_i,) fp32 tensor set to
	zero (the branch's bias gradient).  One dispatch + one combine for all problems.  Returns False (nothing launched) outside the envelope."""
	n = len(xs)
	B, _, T = xs[0].shape
	dt = xs[0].dtype
	cins, couts = [x.shape[1] for x in xs], [dy.shape[1] for dy in dys]
	if n == 0 or n > 12 or dt not in HALF_DTYPES or any(c % 128 for c in cins + couts) or any(not is_cl(t) or t.dtype != dt or t.shape[0] != B or t.shape[2] != T for t in list(xs) + list(dys)):
		return False
	for dw, ci, co in zip(dws, cins, couts):
		assert dw.dtype == torch.float32 and tuple(dw.shape) == (co, ci, 1) and (dw.is_contiguous() or (dw.stride(0) == ci and dw.stride(1) == 1)), (dw.shape, dw.stride())
	key = (tuple(cins), tuple(couts), B, T)
	nb = _wgrad_group_ws.get(key)
	if nb is None:
		nb = _wgrad_group_ws[key] = _lib.load().convasr_wgrad1x1_grouped_workspace_bytes(n, _int_array(cins), _int_array(couts), B, T)
	ws = workspace(nb, xs[0].device, 'wgrad')
	_lib.timed('conv1d_wgrad', 2.0 * B * T * sum(ci * co for ci, co in zip(cins, couts)), lambda: call('convasr_wgrad1x1_grouped', n, _ptr_array(xs), _ptr_array(dys), _ptr_array(dws), _ptr_array(zeros) if zeros else None,
		_int_array(cins), _int_array(couts), _int_array(accumulate) if accumulate else None, ptr(ws), dtype_code(dt), B, T, stream_ptr()))
	return True


def add16(a, b, out = None):
	"""out = a + b for two 16-bit tensors of the same (channels-last) layout; in place when out is a or b."""
	assert a.dtype == b.dtype and a.dtype in HALF_DTYPES and a.shape == b.shape and a.stride() == b.stride() and a.numel() % 8 == 0
	out = torch.empty_like(a) if out is None else out
	call('convasr_add16', ptr(a), ptr(b), ptr(out), a.numel(), dtype_code(a.dtype), stream_ptr())
	return out


_workspaces = {}
_capturing = [False]  # functional.CAPTURING (the same list object, installed by functional at import)


def workspace(nbytes, device, tag = 'default'):
	"""Grow-only scratch buffer per (device, tag, current stream): kernels that share one are ordered by that stream, and a
	buffer replaced by a bigger one is released by the caching allocator in the order of the stream it was allocated on (the
	side-stream wgrad and a main-stream wgrad of the same backward never share or free each other's slabs)."""
	if _capturing[0] or (tag == 'splitk' and device.type == 'cuda' and torch.cuda.is_current_stream_capturing()):  # (the inference path is captured by its callers -- bench_infer.py, a server -- not by train.GraphedTrainStep: ask the runtime)
		return torch.empty(max(int(nbytes), 256), dtype = torch.uint8, device = device)  # inside a graph capture: from the graph's own pool, never a cached buffer a later eager step could replace
	key = (device, tag, _lib.stream_ptr() if device.type == 'cuda' else 0)  # (the current device's current stream: kernels are launched there)
	buf = _workspaces.get(key)
	if buf is None or buf.numel() < nbytes:
		buf = torch.empty(max(int(nbytes), 1 << 20), dtype = torch.uint8, device = device)
		_workspaces[key] = buf
	return buf


def fold2_geometry(K, pad):
	"""(K', P') of the stride-1 conv over the (T / 2, 2 Cin) view that equals a stride-2 conv (K, pad); include/convasr_hip.h."""
	import ctypes
	kf, pf = ctypes.c_int(0), ctypes.c_int(0)
	call('convasr_fold2_geometry', K, pad, ctypes.byref(kf), ctypes.byref(pf))
	return kf.value, pf.value


def fold2_pack_weight(w, dtype, pad, out = None):
	"""(Cout, Cin, K) fp32 parameter -> packed forward operand [K'][cout_pad][2 Cin] of the folded conv (refreshed in place if given)."""
	require_cuda(w)
	w = w.detach()
	layout = weight_layout(w) if w.dtype == torch.float32 else None
	if layout is None:
		w, layout = w.float().contiguous(), _lib.W_REFERENCE
	Cout, Cin, K = w.shape
	if out is None:
		out = torch.empty(fold2_geometry(K, pad)[0], cout_pad(Cout), 2 * Cin, dtype = dtype, device = w.device)
	call('convasr_fold2_pack_weight', ptr(w), layout, ptr(out), dtype_code(dtype), Cout, Cin, K, pad, stream_ptr())
	return out


def fold2_unfold_wgrad(dwf, dw, pad, accumulate = False):
	"""dw (Cout, Cin, K) fp32 (+)= the folded conv's gradient dwf [K'][Cout][2 Cin]."""
	Cout, Cin, K = dw.shape
	layout = weight_layout(dw)
	assert layout is not None and dw.dtype == torch.float32 and dwf.dtype == torch.float32 and dwf.is_contiguous() and tuple(dwf.shape) == (fold2_geometry(K, pad)[0], Cout, 2 * Cin)
	call('convasr_fold2_unfold_wgrad', ptr(dwf), ptr(dw), layout, Cout, Cin, K, pad, int(accumulate), stream_ptr())
	return dw


# ------------------------------------------------------------------------------------------------ split-operand ("x3") convs

SPLIT_INPUT, SPLIT_GRAD = 0, 1  # plane orders of split3(): (hi, lo, hi) for a conv input, (hi, hi, lo) for an output gradient


def split3(x, dtype, order):
	"""fp32 channels-last (B, C, T) -> its three 16-bit planes per frame, memory [B][T][3][C] (include/convasr_hip.h, "split-operand
	convs"), returned as the channels-last (B, 3 C, T) tensor the forward / dgrad kernels read; split3_frames() is the weight gradient's view."""
	B, C, T = x.shape
	assert is_cl(x) and x.dtype == torch.float32 and dtype in HALF_DTYPES and C % 8 == 0
	out = empty_cl(B, 3 * C, T, dtype, x.device)
	_lib.timed('hbm:split3_kernel', 0.0, lambda: call('convasr_split3', ptr(x), ptr(out), dtype_code(dtype), B * T, C, int(order), stream_ptr()), nbytes = float(B * T * C * 10))
	return out


def split3_frames(x3):
	"""The (B, 3 C, T) plane tensor of split3() read as (B, C, 3 T): frame 3 t + p = plane p of frame t (the same memory)."""
	B, C3, T = x3.shape
	assert is_cl(x3) and C3 % 3 == 0
	return x3.as_strided((B, C3 // 3, 3 * T), (T * C3, 1, C3 // 3))


def split3_plane(x3, plane = 0):
	"""One plane of a (B, 3 C, T) plane tensor as a dense channels-last (B, C, T) tensor (a strided copy)."""
	B, C3, T = x3.shape
	C = C3 // 3
	assert is_cl(x3) and C3 % 3 == 0
	out = empty_cl(B, C, T, x3.dtype, x3.device)
	out.permute(0, 2, 1).copy_(x3.as_strided((B, T, C), (T * C3, C3, 1), x3.storage_offset() + plane * C))
	return out


def conv1d_wgrad_hi(x3, dy, Cout, K, dil, pad, dw, accumulate = False, work = None, family = None):
	"""conv1d_wgrad (stride 1) of plane 0 (hi) of the split-operand plane tensor x3 (B, 3 Cin, Tin) against a dense 16-bit dy: the plane is read in
	place (frames 3 Cin elements apart, convasr_conv1d_wgrad_ld) inside the LDS-DMA kernel's envelope, copied out first outside it."""
	B, C3, Tin = x3.shape
	Cin, Tout = C3 // 3, dy.shape[2]
	layout = weight_layout(dw)
	assert is_cl(x3) and is_cl(dy) and x3.dtype == dy.dtype and x3.dtype in HALF_DTYPES and layout is not None and dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, Cin, K), (dw.shape, dw.stride())
	skey = (x3.dtype, B, Cin, Cout, Tin, Tout, K, dil)
	ok = _wgrad_ld_ok.get(skey)
	if ok is None:
		ok = _wgrad_ld_ok[skey] = bool(_lib.load().convasr_conv1d_wgrad_ld_supported(dtype_code(x3.dtype), B, Cin, Cout, Tin, Tout, K, dil, C3, Cout))
	if not ok:  # outside the LDS-DMA kernel's envelope (channel counts not multiples of 128, a tap window too wide for its LDS ring): a dense copy of the plane
		return conv1d_wgrad(split3_plane(x3), dy, Cout, K, 1, dil, pad, dw, accumulate = accumulate, work = work, family = family)
	if K == 1 and (Cout * Cin) % 4 == 0:
		layout = _lib.W_KMAJOR
	wkey = (B, Cin, Cout, Tin, Tout, K, 1, dil)
	nbytes = _wgrad_ws_bytes.get(wkey)
	if nbytes is None:
		nbytes = _wgrad_ws_bytes[wkey] = _lib.load().convasr_conv1d_wgrad_workspace_bytes(*wkey)
	ws = workspace(nbytes, x3.device, 'wgrad')
	_lib.timed(family or 'conv1d_wgrad', 2.0 * B * Tout * Cout * Cin * K if work is None else work, lambda: call('convasr_conv1d_wgrad_ld', ptr(x3), C3, ptr(dy), Cout, ptr(dw), ptr(ws), dtype_code(dy.dtype), B, Cin, Cout, Tin, Tout, K, dil, pad, int(accumulate), layout, stream_ptr()))
	return dw


_wgrad_ld_ok = {}


def pack_weight_split3(w, dtype, out = None, want_dgrad = True, dgrad_planes = 3):
	"""(Cout, Cin, K) fp32 parameter -> (fwd [K][cout_pad][3 Cin], dgrad [K][cin_pad][3 Cout]) split operands, refreshed in place when
	`out` = (fwd, dgrad) of an earlier call.  want_dgrad = False: the forward planes only (dgrad is returned as None).  dgrad_planes = 1: the
	dgrad operand as the ordinary 16-bit one, [K][cin_pad][Cout] (w_hi alone: a one-product backward)."""
	require_cuda(w)
	w = w.detach()
	layout = weight_layout(w) if w.dtype == torch.float32 else None
	if layout is None:
		w, layout = w.float().contiguous(), _lib.W_REFERENCE
	Cout, Cin, K = w.shape
	fwd, dgr = out if out is not None else (None, None)
	if fwd is None:
		fwd = torch.zeros(K, cout_pad(Cout), 3 * Cin, dtype = dtype, device = w.device)
	if dgr is None and want_dgrad:
		dgr = torch.zeros(K, cout_pad(Cin), dgrad_planes * Cout, dtype = dtype, device = w.device)
	assert dgr is None or dgr.shape[2] == dgrad_planes * Cout
	call('convasr_pack_conv_weight_split3', ptr(w), layout, ptr(fwd), ptr(dgr) if want_dgrad else None, dgrad_planes, dtype_code(dtype), Cout, Cin, K, stream_ptr())
	return fwd, (dgr if want_dgrad else None)


def colsum(y, out, accumulate = False):
	"""out (C,) fp32 (+)= sum over (b, t) of a channels-last (B, C, T) tensor: a conv's bias gradient on its own."""
	B, C, T = y.shape
	assert is_cl(y) and out.dtype == torch.float32 and out.numel() == C and out.is_contiguous()
	ws = workspace(_lib.load().convasr_colsum_workspace_bytes(B * T, C), y.device, 'colsum')
	call('convasr_colsum', ptr(y), dtype_code(y.dtype), B * T, C, ptr(out), ptr(ws), int(accumulate), stream_ptr())
	return out


_wgrad_ws_bytes = {}


def conv1d_wgrad(x, dy, Cout, K, stride, dil, pad, dw, dbias = None, accumulate = False, work = None, family = None):
	"""dw (Cout, Cin, K) fp32 (+)= wgrad; x, dy channels-last of the same dtype.  dw is torch-contiguous (the reference's layout) or a
	view of tap-major memory (weight_layout: the training arena's gradients)."""
	B, Cin, Tin = x.shape
	Tout = dy.shape[2]
	layout = weight_layout(dw)
	assert is_cl(x) and is_cl(dy) and x.dtype == dy.dtype and layout is not None and dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, Cin, K), (dw.shape, dw.stride())
	if K == 1 and (Cout * Cin) % 4 == 0:
		layout = _lib.W_KMAJOR  # one tap: the two layouts are the same memory, and the tap-major combine is the streaming one
	wkey = (B, Cin, Cout, Tin, Tout, K, stride, dil)
	nbytes = _wgrad_ws_bytes.get(wkey)
	if nbytes is None:
		nbytes = _wgrad_ws_bytes[wkey] = _lib.load().convasr_conv1d_wgrad_workspace_bytes(*wkey)
	ws = workspace(nbytes, x.device, 'wgrad')
	_lib.timed(family or 'conv1d_wgrad', 2.0 * B * Tout * Cout * Cin * K if work is None else work, lambda: call('convasr_conv1d_wgrad', ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(ws), dtype_code(x.dtype), B, Cin, Cout, Tin, Tout, K, stride, dil, pad, int(accumulate), layout, stream_ptr()))
	return dw


# ------------------------------------------------------------------------------------------------ grouped conv (the separable block's first half)

def grouped_conv1d(x, w, bias, groups, stride = 1, pad = 0, relu = True):
	"""nn.Conv1d(Cin, Cout, K, groups = G) + bias (+ ReLU) on a channels-last activation; w: fp32 (Cout, Cin / G, K) of any strides."""
	B, Cin, Tin = x.shape
	Cout, cgi, K = w.shape
	assert is_cl(x) and w.dtype == torch.float32 and cgi * groups == Cin
	Tout = conv_out_len(Tin, K, stride, 1, pad)
	y = empty_cl(B, Cout, Tout, x.dtype, x.device)
	call('convasr_grouped_conv1d_fwd', ptr(x), ptr(w), w.stride(0), w.stride(1), w.stride(2), ptr(bias), ptr(y), dtype_code(x.dtype), B, Cin, Cout, Tin, Tout, K, stride, pad, groups, int(relu), stream_ptr())
	return y


def grouped_conv1d_dgrad(dy, y_act, w, Cin, Tin, groups, stride = 1, pad = 0):
	B, Cout, Tout = dy.shape
	K = w.shape[2]
	assert is_cl(dy) and (y_act is None or (is_cl(y_act) and y_act.dtype == dy.dtype))
	dx = empty_cl(B, Cin, Tin, dy.dtype, dy.device)
	call('convasr_grouped_conv1d_dgrad', ptr(dy), ptr(y_act), ptr(w), w.stride(0), w.stride(1), w.stride(2), ptr(dx), dtype_code(dy.dtype), B, Cin, Cout, Tin, Tout, K, stride, pad, groups, stream_ptr())
	return dx


def grouped_conv1d_wgrad(x, dy, y_act, dw, dbias, groups, stride = 1, pad = 0, accumulate = False):
	B, Cin, Tin = x.shape
	Cout, Tout, K = dy.shape[1], dy.shape[2], dw.shape[2]
	assert is_cl(x) and is_cl(dy) and dw.dtype == torch.float32 and tuple(dw.shape) == (Cout, Cin // groups, K)
	ws = workspace(_lib.load().convasr_grouped_conv1d_wgrad_workspace_bytes(B, Cin, Cout, K, groups), x.device, 'grouped_wgrad')
	call('convasr_grouped_conv1d_wgrad', ptr(x), ptr(dy), ptr(y_act), ptr(dw), dw.stride(0), dw.stride(1), dw.stride(2), ptr(dbias), ptr(ws), dtype_code(x.dtype), B, Cin, Cout, Tin, Tout, K, stride, pad, groups, int(accumulate), stream_ptr())
	return dw


# ------------------------------------------------------------------------------------------------ batch norm + activation

def bn_finalize(stats, n, gamma, beta, running_mean, running_var, momentum, eps, num_batches_tracked = None):
	"""stats: a ConvStats (partial rows of the conv launch) or a (2 C,) fp64 tensor of totals."""
	if isinstance(stats, ConvStats):
		buf, rows, C = stats.buf, stats.rows, stats.C
	else:
		buf, rows, C = stats, 1, stats.numel() // 2
	out = torch.empty(4, C, dtype = torch.float32, device = buf.device)  # mean, invstd, scale, shift
	o = out.data_ptr()
	call('convasr_bn_finalize', ptr(buf), rows, n, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(momentum), float(eps), o, o + 4 * C, o + 8 * C, o + 12 * C, C, ptr(num_batches_tracked), stream_ptr())
	return out


def _float_array(vals):
	return (ctypes.c_float * max(len(vals), 1))(*[float(v) for v in vals])


def bn_finalize_grouped(stats, n, gammas, betas, running_means, running_vars, momenta, epss, num_batches_tracked):
	"""bn_finalize for several batch norms of one channel count in one launch; stats: ConvStats of equal row counts.  Returns one (4, C)
	fp32 tensor per batch norm (mean, invstd, scale, shift)."""
	k, C, rows = len(stats), stats[0].C, stats[0].rows
	assert all(s_.C == C and s_.rows == rows for s_ in stats)
	out = torch.empty(k, 4, C, dtype = torch.float32, device = stats[0].buf.device)
	call('convasr_bn_finalize_grouped', k, _ptr_array([s_.buf for s_ in stats]), rows, n, _ptr_array(gammas), _ptr_array(betas), _ptr_array(running_means), _ptr_array(running_vars),
		_float_array(momenta), _float_array(epss), _ptr_array([out[i] for i in range(k)]), _ptr_array(num_batches_tracked), C, stream_ptr())
	return [out[i] for i in range(k)]


def bn_bwd_finalize_grouped(sums, gammas, means, invstds, n, coefs, dgammas, dbetas, accumulate):
	"""bn_bwd_finalize for several batch norms in one launch; sums: (2 C,) fp64 totals each."""
	k, C = len(sums), sums[0].numel() // 2
	call('convasr_bn_bwd_finalize_grouped', k, _ptr_array(sums), _int_array([1] * k), _ptr_array(gammas), _ptr_array(means), _ptr_array(invstds), _ptr_array(coefs), _ptr_array(dgammas), _ptr_array(dbetas),
		_int_array(accumulate), int(n), C, stream_ptr())


def bn_bwd_reduce_many(dz, gate, dropout_p, ys, means, invstds, gammas, coefs, dgammas, dbetas, accumulate):
	"""Pass 1 of a dense block's backward in one sweep from the stored gates (include/convasr_hip.h): returns g; fills coefs / dgammas / dbetas."""
	B, C, T = dz.shape
	assert is_cl(dz) and dz.dtype in HALF_DTYPES and all(is_cl(y) and y.dtype == dz.dtype and y.shape == dz.shape for y in ys) and gate.dtype == torch.uint8 and gate.numel() * 8 == dz.numel()
	g = empty_cl(B, C, T, dz.dtype, dz.device)
	k = len(ys)
	ws = workspace(_lib.load().convasr_bn_bwd_reduce_many_workspace_bytes(k, B, T, C), dz.device, 'bn_bwd')
	_lib.timed('hbm:bn_act_bwd_reduce_kernel', 0.0, lambda: call('convasr_bn_bwd_reduce_many', ptr(dz), ptr(gate), float(dropout_p), ptr(g), k, _ptr_array(ys), _ptr_array(means), _ptr_array(invstds), _ptr_array(gammas),
		_ptr_array(coefs), _ptr_array(dgammas), _ptr_array(dbetas), _int_array(accumulate), ptr(ws), dtype_code(dz.dtype), B, T, C, stream_ptr()), nbytes = float(B * T * C * dz.element_size() * (2 + k)))
	return g


def bn_bwd_apply_grouped(g, ys, coefs):
	"""dy_i = A_i g + B_i y_i + D_i for every (y_i, coef_i), g read once.  Returns the list of dy_i."""
	B, C, T = g.shape
	assert is_cl(g) and g.dtype in HALF_DTYPES and all(is_cl(y) and y.dtype == g.dtype and y.shape == g.shape for y in ys)
	dys = [empty_cl(B, C, T, g.dtype, g.device) for _ in ys]
	_lib.timed('hbm:bn_act_bwd_apply_kernel', 0.0, lambda: call('convasr_bn_bwd_apply_grouped', ptr(g), len(ys), _ptr_array(ys), _ptr_array(coefs), _ptr_array(dys), dtype_code(g.dtype), B, T, C, stream_ptr()), nbytes = float(B * T * C * g.element_size() * (1 + 2 * len(ys))))
	return dys


def bn_eval_scale_shift(gamma, beta, running_mean, running_var, eps):
	C = running_mean.numel()
	out = torch.empty(2, C, dtype = torch.float32, device = running_mean.device)
	call('convasr_bn_eval_scale_shift', ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), float(eps), ptr(out[0]), ptr(out[1]), C, stream_ptr())
	return out


def _ptr_array(items):
	import ctypes
	arr = (ctypes.c_void_p * max(len(items), 1))()
	for i, t in enumerate(items):
		arr[i] = None if t is None else t.data_ptr()
	return arr


def bn_act(y, scale, shift, act, xlen = None, res = (), rscale = (), rshift = (), dropout_p = 0.0, seed = 0, offset = 0, out = None, gate = None, step_key = None, planes = None):
	"""gate (optional uint8 (B*T*C/8,)): receives one bit per element, set iff the gradient passes it (include/convasr_hip.h).
	step_key (optional, here and in the backward passes): device address (int) of the per-step dropout key word (functional.begin_step).
	planes (a 16-bit dtype; fp32 y with scale / shift and a clamp-type activation): the result is returned as its split-operand planes, the
	(B, 3 C, T) tensor split3(z, planes, SPLIT_INPUT) would produce, written by this pass itself."""
	B, C, T = y.shape
	assert gate is None or (gate.dtype == torch.uint8 and gate.numel() * 8 == B * C * T and gate.is_contiguous())
	assert is_cl(y) and all(is_cl(r) and r.dtype == y.dtype for r in res)
	if planes is not None:
		assert y.dtype == torch.float32 and planes in HALF_DTYPES and out is None and scale is not None and act[0] != _lib.ACT_LEAKY_RELU
		z3 = empty_cl(B, 3 * C, T, planes, y.device)
		_lib.timed('hbm:bn_act_fwd_kernel', 0.0, lambda: call('convasr_bn_act_fwd_split3', ptr(y), ptr(z3), dtype_code(planes), ptr(scale), ptr(shift), len(res), _ptr_array(res), _ptr_array(rscale) if rscale else None, _ptr_array(rshift) if rshift else None, act[0], act[1], act[2], float(dropout_p), int(seed), int(offset), step_key, ptr(xlen), B, T, C, ptr(gate), stream_ptr()), nbytes = float(B * T * C * (10 + 4 * len(res))))
		return z3
	z = out if out is not None else empty_cl(B, C, T, y.dtype, y.device)
	_lib.timed('hbm:bn_act_fwd_kernel', 0.0, lambda: call('convasr_bn_act_fwd', ptr(y), ptr(z), dtype_code(y.dtype), ptr(scale), ptr(shift), len(res), _ptr_array(res), _ptr_array(rscale) if rscale else None, _ptr_array(rshift) if rshift else None, act[0], act[1], act[2], float(dropout_p), int(seed), int(offset), step_key, ptr(xlen), B, T, C, ptr(gate), stream_ptr()), nbytes = float(B * T * C * y.element_size() * (2 + len(res))))
	return z


_bn_bwd_ws_bytes = {}


def bn_act_bwd_reduce(dz, y, scale, shift, mean, invstd, act, xlen = None, res = (), rscale = (), rshift = (), rmean = (), rinvstd = (), rsums = (), dropout_p = 0.0, seed = 0, offset = 0, sums = None, write_g = True, gamma = None, coef = None, dgamma = None, dbeta = None, accumulate = False, gate = None, step_key = None):
	"""gate: the forward pass's one-bit gradient gates (bn_act(..., gate = ...)); needs write_g = False and no residuals."""
	B, C, T = y.shape
	assert is_cl(y) and is_cl(dz) and dz.dtype == y.dtype and (gate is None or (not write_g and not res))
	g = empty_cl(B, C, T, y.dtype, y.device) if write_g else None
	nb = _bn_bwd_ws_bytes.get((B, T, C))
	if nb is None:
		nb = _bn_bwd_ws_bytes[(B, T, C)] = _lib.load().convasr_bn_bwd_workspace_bytes(B, T, C)
	ws = workspace(nb, y.device, 'bn_bwd')
	_lib.timed('hbm:bn_act_bwd_reduce_kernel', 0.0, lambda: call('convasr_bn_act_bwd_reduce', ptr(dz), ptr(y), ptr(g), dtype_code(y.dtype), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), len(res), _ptr_array(res), _ptr_array(rscale) if rscale else None, _ptr_array(rshift) if rshift else None, _ptr_array(rmean) if rmean else None, _ptr_array(rinvstd) if rinvstd else None, _ptr_array(rsums) if rsums else None, act[0], act[1], act[2], float(dropout_p), int(seed), int(offset), step_key, ptr(xlen), ptr(sums), ptr(ws), ptr(gamma), ptr(coef), ptr(dgamma), ptr(dbeta), int(accumulate), B, T, C, ptr(gate), stream_ptr()), nbytes = float(B * T * C * y.element_size() * (2 + len(res) + (1 if write_g else 0))))
	return g


def bn_act_bwd_apply(dz_or_g, y, coef, from_dz, scale = None, shift = None, act = (_lib.ACT_NONE, 0.0, 0.0), xlen = None, dropout_p = 0.0, seed = 0, offset = 0, out = None, gate = None, step_key = None, planes = None, hi_only = False):
	"""planes (a 16-bit dtype, fp32 inputs only): dy is returned as its split-operand planes, the (B, 3 C, T) tensor split3(dy, planes, SPLIT_GRAD)
	would produce -- written by this pass itself; hi_only: as the dense (B, C, T) tensor of that type instead (dy rounded once: the operand of a
	one-product backward)."""
	B, C, T = y.shape
	if planes is not None and hi_only:
		assert y.dtype == torch.float32 and planes in HALF_DTYPES and out is None
		dy16 = empty_cl(B, C, T, planes, y.device)
		_lib.timed('hbm:bn_act_bwd_apply_kernel', 0.0, lambda: call('convasr_bn_act_bwd_apply_to_half', ptr(dz_or_g), ptr(y), ptr(dy16), dtype_code(planes), ptr(coef), int(from_dz), ptr(scale), ptr(shift), act[0], act[1], act[2], float(dropout_p), int(seed), int(offset), step_key, ptr(xlen), B, T, C, ptr(gate), stream_ptr()), nbytes = float(B * T * C * 10))
		return dy16
	if planes is not None:
		assert y.dtype == torch.float32 and planes in HALF_DTYPES and out is None
		dy3 = empty_cl(B, 3 * C, T, planes, y.device)
		_lib.timed('hbm:bn_act_bwd_apply_kernel', 0.0, lambda: call('convasr_bn_act_bwd_apply_split3', ptr(dz_or_g), ptr(y), ptr(dy3), dtype_code(planes), ptr(coef), int(from_dz), ptr(scale), ptr(shift), act[0], act[1], act[2], float(dropout_p), int(seed), int(offset), step_key, ptr(xlen), B, T, C, ptr(gate), stream_ptr()), nbytes = float(B * T * C * 14))
		return dy3
	dy = out if out is not None else empty_cl(B, C, T, y.dtype, y.device)
	_lib.timed('hbm:bn_act_bwd_apply_kernel', 0.0, lambda: call('convasr_bn_act_bwd_apply', ptr(dz_or_g), ptr(y), ptr(dy), dtype_code(y.dtype), ptr(coef), int(from_dz), ptr(scale), ptr(shift), act[0], act[1], act[2], float(dropout_p), int(seed), int(offset), step_key, ptr(xlen), B, T, C, ptr(gate), stream_ptr()), nbytes = float(B * T * C * y.element_size() * 3))
	return dy


def bn_bwd_apply(g, y, gamma, mean, invstd, sums, dgamma = None, dbeta = None, accumulate = False, need_dy = True, inplace = True):
	B, C, T = y.shape
	dy = (g if inplace else empty_cl(B, C, T, y.dtype, y.device)) if need_dy else None
	call('convasr_bn_bwd_apply', ptr(g), ptr(y), ptr(dy), dtype_code(y.dtype), ptr(gamma), ptr(mean), ptr(invstd), ptr(sums), ptr(dgamma), ptr(dbeta), int(accumulate), B, T, C, stream_ptr())
	return dy


# ------------------------------------------------------------------------------------------------ head

def log_softmax(logits):
	"""channels-last fp32 (B, C, T) -> same."""
	B, C, T = logits.shape
	assert is_cl(logits) and logits.dtype == torch.float32
	out = empty_cl(B, C, T, torch.float32, logits.device)
	call('convasr_log_softmax_fwd', ptr(logits), ptr(out), B * T, C, stream_ptr())
	return out


def log_softmax_bwd(grad_lp, log_probs):
	B, C, T = log_probs.shape
	grad_lp = as_cl(grad_lp, torch.float32)
	out = empty_cl(B, C, T, torch.float32, log_probs.device)
	call('convasr_log_softmax_bwd', ptr(grad_lp), ptr(log_probs), ptr(out), B * T, C, stream_ptr())
	return out


def ctc_loss(log_probs, targets, olen, ylen, blank, need_grad = True):
	"""log_probs channels-last fp32 (B, C, T); returns (nll (B,), grad channels-last (B, C, T) or None)."""
	B, C, T = log_probs.shape
	assert is_cl(log_probs) and log_probs.dtype == torch.float32
	dev = log_probs.device
	targets = targets.to(device = dev, dtype = torch.int64).contiguous()
	if targets.ndim == 1:
		targets = targets.view(B, -1)
	olen = olen.to(device = dev, dtype = torch.int64).contiguous()
	ylen = ylen.to(device = dev, dtype = torch.int64).contiguous()
	S_max = targets.shape[1]
	nbytes = _lib.load().convasr_ctc_workspace_bytes(B, T, S_max)
	if nbytes < 0:
		raise _lib.ConvasrHipError(f'ctc_loss: target length {S_max} unsupported')
	ws = workspace(nbytes, dev, 'ctc')
	nll = torch.empty(B, dtype = torch.float32, device = dev)
	grad = empty_cl(B, C, T, torch.float32, dev) if need_grad else None
	call('convasr_ctc_loss', ptr(log_probs), ptr(targets), ptr(olen), ptr(ylen), ptr(nll), ptr(grad), ptr(ws), B, T, C, S_max, blank, stream_ptr())
	return nll, grad


def scale_rows(grad, gscale, gdiv = None):
	"""out[b] = grad[b] * gscale[b] (/ gdiv[b] if given: an int64 (B,) vector, possibly a strided column); gscale None: grad[b] / gdiv[b]."""
	B = grad.shape[0]
	out = torch.empty_like(grad)  # preserves strides (channels-last stays channels-last)
	assert gdiv is None or (gdiv.dtype == torch.int64 and gdiv.ndim == 1 and gdiv.shape[0] == B and gdiv.device == grad.device)
	call('convasr_scale_rows', ptr(grad), None if gscale is None else ptr(gscale.to(torch.float32).contiguous()), ptr(gdiv), 0 if gdiv is None else gdiv.stride(0), ptr(out), B, grad.numel() // B, stream_ptr())
	return out


def loss_head(loss_vec, ylen_col, ent = None, accumulate_iterations = 1, need_grad = True, loss_scaler = None, metric_scale = 1.0):
	"""train.py:754-756 + the gate of 769 in one launch.  Returns (out3 = [loss, loss_cur, entropy] fp32, grad_loss_vec (B,) or None,
	skipped: 1-element bool).  loss_scaler: the current state of a dynamic loss scaler (fp16 training): grad_loss_vec is scaled by it;
	metric_scale: factor on the two logged means (1 / world size ahead of a SUM all-reduce)."""
	require_cuda(loss_vec)
	B = loss_vec.shape[0]
	lv = loss_vec.detach().to(torch.float32).contiguous()
	assert ylen_col.dtype == torch.int64 and ylen_col.ndim == 1 and ylen_col.shape[0] == B and ylen_col.device == lv.device
	out3 = torch.empty(3, dtype = torch.float32, device = lv.device)
	gvec = torch.empty(B, dtype = torch.float32, device = lv.device) if need_grad else None
	skipped = torch.empty(1, dtype = torch.bool, device = lv.device)
	ent = None if ent is None else ent.detach().to(torch.float32).contiguous()
	call('convasr_loss_head', ptr(lv), ptr(ylen_col), ylen_col.stride(0), ptr(ent), B, float(accumulate_iterations), ptr(out3), ptr(gvec), ptr(skipped), ptr(loss_scaler), float(metric_scale), stream_ptr())
	return out3, gvec, skipped


def entropy(log_probs, olen = None, eps = 1e-9):
	B, C, T = log_probs.shape
	lp = as_cl(log_probs, torch.float32)
	ent = torch.empty(B, dtype = torch.float32, device = lp.device)
	ol = None if olen is None else olen.to(device = lp.device, dtype = torch.int64).contiguous()
	call('convasr_entropy', ptr(lp), ptr(ol), ptr(ent), B, T, C, float(eps), stream_ptr())
	return ent


def weighted_mean_entropy(log_probs, olen = None, eps = 1e-9, eps_id = -1):
	B, C, T = log_probs.shape
	lp = as_cl(log_probs, torch.float32)
	out = torch.empty(B, dtype = torch.float32, device = lp.device)
	ol = None if olen is None else olen.to(device = lp.device, dtype = torch.int64).contiguous()
	call('convasr_weighted_mean_entropy', ptr(lp), ptr(ol), ptr(out), B, T, C, int(eps_id) % C, float(eps), stream_ptr())
	return out


def normalize_signal(signal, eps = 1e-5, denom_multiplier = 1.0):
	"""models.py:684-686 as a standalone op: per-utterance peak (absmax kernel), then one scaling pass."""
	require_cuda(signal)
	assert signal.ndim == 2
	if signal.numel() == 0:
		return signal
	signal = signal.contiguous() if signal.dtype in (torch.float32, torch.int16) else signal.float().contiguous()
	B, T = signal.shape
	absmax = torch.empty(B, dtype = torch.float32, device = signal.device)
	call('convasr_signal_absmax', ptr(signal), dtype_code(signal.dtype), B, T, ptr(absmax), stream_ptr())
	x = signal if signal.dtype == torch.float32 else signal.float()
	out = torch.empty_like(x)
	scale = ((absmax + eps) * denom_multiplier).reciprocal()
	call('convasr_scale_rows', ptr(x), ptr(scale), None, 0, ptr(out), B, T, stream_ptr())
	return out


def argmax(log_probs):
	B, C, T = log_probs.shape
	lp = as_cl(log_probs, torch.float32)
	idx = torch.empty(B, T, dtype = torch.int64, device = lp.device)
	call('convasr_argmax', ptr(lp), ptr(idx), B * T, C, stream_ptr())
	return idx


# ------------------------------------------------------------------------------------------------ optimizer

def sumsq(flat_grad, out = None, norm_out = None, norm_scale = 1.0, loss_scaler = None):
	"""out[0] = sum of squares (fp64); norm_out (1-element fp32, optional) = sqrt(out) * norm_scale (/ the loss scaler's scale)."""
	out = out if out is not None else torch.empty(1, dtype = torch.float64, device = flat_grad.device)
	ws = workspace(_lib.load().convasr_sumsq_workspace_bytes(), flat_grad.device, 'sumsq')
	call('convasr_sumsq', ptr(flat_grad), flat_grad.numel(), ptr(out), ptr(ws), ptr(norm_out), float(norm_scale), ptr(loss_scaler), stream_ptr())
	return out


def _scaler_pair(scaler):
	"""(state read by this step, state written by it) of a dynamic loss scaler, or (None, None)."""
	if scaler is None:
		return None, None
	s_in, s_out = scaler
	assert s_in.dtype == s_out.dtype == torch.float32 and s_in.numel() == s_out.numel() == _lib.LOSS_SCALER_FLOATS and s_in.data_ptr() != s_out.data_ptr()
	return s_in, s_out


def sgd_step(p, g, buf, n, sumsq_buf, max_norm, lr, momentum, weight_decay, nesterov, first, grad_out = None, loss_gate = None, grad_scale = 1.0, p16 = None, scaler = None, lr_dev = None):
	"""lr_dev: optional 1-element fp32 device tensor that replaces `lr` when the kernel runs (captured step graphs); p16: optional 16-bit mirror of the parameters (bf16 or fp16, n elements); scaler: optional (state_in, state_out) of a dynamic loss scaler."""
	assert loss_gate is None or (loss_gate.dtype == torch.float32 and loss_gate.numel() == 1)
	assert p16 is None or (p16.dtype in HALF_DTYPES and p16.numel() == n)
	s_in, s_out = _scaler_pair(scaler)
	call('convasr_sgd_step', ptr(p), ptr(g), ptr(buf), ptr(grad_out), n, ptr(sumsq_buf), float(max_norm), float(lr), float(momentum), float(weight_decay), int(nesterov), int(first), ptr(loss_gate), float(grad_scale), ptr(p16), _lib.BF16 if p16 is None else dtype_code(p16.dtype), ptr(s_in), ptr(s_out), ptr(lr_dev), stream_ptr())


def adamw_step(p, g, exp_avg, exp_avg_sq, n, sumsq_buf, max_norm, lr, beta1, beta2, eps, weight_decay, step_in, step_out, loss_gate = None, grad_scale = 1.0, p16 = None, scaler = None, lr_dev = None):
	"""One fused torch.optim.AdamW step (+ clip_grad_norm_) over the flat arena; step_in / step_out: 1-element fp32 device tensors (applied-step counter)."""
	assert loss_gate is None or (loss_gate.dtype == torch.float32 and loss_gate.numel() == 1)
	assert p16 is None or (p16.dtype in HALF_DTYPES and p16.numel() == n)
	assert step_in.dtype == step_out.dtype == torch.float32 and step_in.data_ptr() != step_out.data_ptr()
	s_in, s_out = _scaler_pair(scaler)
	call('convasr_adamw_step', ptr(p), ptr(g), ptr(exp_avg), ptr(exp_avg_sq), n, ptr(sumsq_buf), float(max_norm), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), ptr(step_in), ptr(step_out), ptr(loss_gate), float(grad_scale), ptr(p16), _lib.BF16 if p16 is None else dtype_code(p16.dtype), ptr(s_in), ptr(s_out), ptr(lr_dev), stream_ptr())


def conv1d_dgrad_bn_reduce(dy, packed_dgrad, Cin, K, dil, pad, bn_y, bn_scale, bn_shift, bn_mean, bn_invstd, act, dropout_p, seed, offset, xlen, bn_sums, work = None, gate = None, step_key = None):
	"""dx = dgrad(dy) with pass 1 of the consumer layer's batch-norm backward fused into the epilogue (bn_sums += per-channel sums).
	Returns dx, or None when the shape is outside the fused kernel's envelope (nothing was launched)."""
	import ctypes
	B, Cout, Tdy = dy.shape
	T = conv_out_len(Tdy, K, 1, dil, pad)
	assert is_cl(dy) and dy.dtype in HALF_DTYPES and is_cl(bn_y) and bn_y.dtype == dy.dtype and tuple(bn_y.shape) == (B, Cin, T) and isinstance(bn_sums, ConvStats) and bn_sums.fits(Cin, B, T, dy.device), (dy.shape, bn_y.shape, Cin, T)
	rows = ctypes.c_int(0)
	dx = empty_cl(B, Cin, T, dy.dtype, dy.device)
	rc = [0]
	def run():
		rc[0] = _lib.call_rc('convasr_conv1d_dgrad_bn_reduce', ptr(dy), ptr(packed_dgrad), ptr(dx), dtype_code(dy.dtype), B, Cout, Cin, Tdy, T, K, dil, pad, ptr(bn_y), ptr(bn_scale), ptr(bn_shift), ptr(bn_mean), ptr(bn_invstd), act[0], act[1], act[2], float(dropout_p), int(seed), int(offset), step_key, ptr(xlen), ptr(bn_sums.buf), ctypes.byref(rows), ptr(gate), stream_ptr())
	family = 'conv1d_igemm_v2s_kernel<bf16>+bn_bwd' if Cout % 64 == 0 else 'conv1d_igemm (other variants)'
	flops, nbytes_ = 2.0 * B * T * Cout * Cin * K if work is None else work, float(B * Tdy * Cout * 2 + K * Cout * Cin * 2 + 2 * B * T * Cin * 2)
	symbol = 'v2s16' if family.startswith('conv1d_igemm_v2s') else None
	if family.startswith('conv1d_igemm_v2s') and memory_bound(flops, nbytes_):
		family = 'hbm:conv1d_igemm_v2s_kernel (memory-bound launches: the 38-class decoder)'
	_lib.timed(family, flops, run, nbytes = nbytes_, symbol = symbol)
	bn_sums.rows = rows.value
	return dx if rc[0] == 0 else None


def bn_bwd_finalize(sums, gamma, mean, invstd, n, coef = None, dgamma = None, dbeta = None, accumulate = False):
	"""sums: the ConvStats the fused dgrad epilogue filled (partial rows of sum g, sum g*xhat), or a (2 C,) fp64 tensor of totals."""
	buf, rows, C = (sums.buf, sums.rows, sums.C) if isinstance(sums, ConvStats) else (sums, 1, sums.numel() // 2)
	call('convasr_bn_bwd_finalize', ptr(buf), rows, ptr(gamma), ptr(mean), ptr(invstd), ptr(coef), ptr(dgamma), ptr(dbeta), int(accumulate), int(n), C, stream_ptr())


# ------------------------------------------------------------------------------------------------ SURVEY 8(f) "next" rows

def novograd_work_table(offsets_host, device):
	"""Static work table of the fused NovoGrad step: every tensor (segment of the arena) cut into items of bounded size.
	Returns (items (n_items, 3) int64, seg_first (n_seg + 1,) int64, item_part (n_items,) fp64 scratch), all on `device`."""
	item = _lib.load().convasr_novograd_item_elems()
	items, seg_first = [], [0]
	for s_, (lo, hi) in enumerate(zip(offsets_host[:-1], offsets_host[1:])):
		for b in range(lo, max(hi, lo + 1), item):
			items.append((s_, b, min(hi, b + item)))
		seg_first.append(len(items))
	return (torch.tensor(items, dtype = torch.int64, device = device), torch.tensor(seg_first, dtype = torch.int64, device = device), torch.empty(len(items), dtype = torch.float64, device = device))


def novograd_step(p, g, mom, ema_in, ema_out, g2, offsets, n, table, max_norm, lr, beta1, beta2, eps, weight_decay, dampening, first, loss_gate = None, total_norm = None, grad_scale = 1.0, p16 = None, scaler = None, lr_dev = None):
	"""One fused NovoGrad step (+ clip_grad_norm_) over the flat arena; offsets: device int64 [n_seg + 1]; table: novograd_work_table(...)."""
	items, seg_first, item_part = table
	assert offsets.dtype == torch.int64 and ema_in.data_ptr() != ema_out.data_ptr() and g2.dtype == torch.float64
	n_seg = offsets.numel() - 1
	s_in, s_out = _scaler_pair(scaler)
	assert p16 is None or (p16.dtype in HALF_DTYPES and p16.numel() == n)
	assert int(first) >= 0 or (ema_in.numel() == n_seg + 1 and ema_out.numel() == n_seg + 1), 'first = -1 (device-side first-step detection) needs the applied-step counter behind the EMAs'
	call('convasr_novograd_step', ptr(p), ptr(g), ptr(mom), ptr(ema_in), ptr(ema_out), ptr(g2), ptr(offsets), n_seg, n, ptr(items), items.shape[0], ptr(seg_first), ptr(item_part), float(max_norm or 0.0), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(bool(dampening)), int(first), ptr(loss_gate), ptr(total_norm), float(grad_scale), ptr(p16), _lib.BF16 if p16 is None else dtype_code(p16.dtype), ptr(s_in), ptr(s_out), ptr(lr_dev), stream_ptr())


def ctc_alignment(log_probs_btc, targets, input_lengths, target_lengths, blank):
	"""log_probs_btc: contiguous (B, T, C) fp32 on the GPU.  Returns (B, S_max) int64 (see include/convasr_hip.h)."""
	B, T, C = log_probs_btc.shape
	dev = log_probs_btc.device
	assert log_probs_btc.is_contiguous() and log_probs_btc.dtype == torch.float32
	targets = targets.to(device = dev, dtype = torch.int64).contiguous()
	il = input_lengths.to(device = dev, dtype = torch.int64).contiguous()
	tl = target_lengths.to(device = dev, dtype = torch.int64).contiguous()
	S_max = targets.shape[1]
	out = torch.empty(B, S_max, dtype = torch.int64, device = dev)
	ws = workspace(_lib.load().convasr_ctc_alignment_workspace_bytes(B, T, S_max), dev, 'ctc_alignment')
	call('convasr_ctc_alignment', ptr(log_probs_btc), ptr(targets), ptr(il), ptr(tl), ptr(out), ptr(ws), B, T, C, S_max, int(blank), stream_ptr())
	return out
