"""MI355X-native mirror of convasr's models.py module surface for the hot path.

Same class names, constructor arguments, forward() signatures, return dictionaries and state-dict keys as the reference
(/root/reference/models.py: LogFilterBankFrontend 486-603, MaskedInstanceNorm1d 688-719, ConvBn1d 80-151, JasperNet 158-347,
Decoder 23-44, configs 819-1442), so `getattr(models, args.model)(...)`, `load_state_dict(checkpoint['model_state_dict'])`
and `model(x, xlen, y = y, ylen = ylen)` work unchanged -- but every arithmetic operation is a hand-written gfx950 kernel
reached through the C ABI in include/convasr_hip.h.  nn.Conv1d / nn.BatchNorm1d objects are used purely as parameter
containers (identical names, shapes and initialisation to the reference); their own forward() is never called.

Activations flow between modules as (B, C, T) tensors whose memory is channels-last (strides (T*C, 1, C)); module outputs
handed back to the caller (logits, log_probs) keep that logical (B, C, T) shape.  There is no CPU implementation here.
"""
import math
import os
import typing
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import ops, _lib
from . import functional as Fn
from .functional import ConvSpec

FP16_TINY = float(torch.finfo(torch.float16).tiny)


# ------------------------------------------------------------------------------------------------ helpers (models.py:611-733)

def compute_output_lengths(x, lengths_fraction = None):
	"""models.py:611-614: valid frames per utterance of x's time axis, int64 (B,) = ceil(fraction * T) evaluated in the fraction's own
	floating type; every utterance is T frames long when no fractions are given.  (The network itself calls ops.output_lengths: one launch.)"""
	B, T = x.shape[0], x.shape[-1]
	if lengths_fraction is None:
		return torch.full((B, ), T, dtype = torch.int64, device = x.device)
	return torch.ceil(lengths_fraction * T).to(torch.int64)


def temporal_mask(x, lengths):
	"""models.py:617-619: bool mask broadcastable to x, True for t < length."""
	return (torch.arange(x.shape[-1], device = x.device, dtype = lengths.dtype).unsqueeze(0) < lengths.unsqueeze(1)).view(x.shape[:1] + (1, ) * (len(x.shape) - 2) + x.shape[-1:])


def entropy(log_probs, lengths = None, dim = 1, eps = 1e-9, sum = True, keepdim = False):
	"""models.py:645-657 (the logged training metric): per-utterance mean entropy over valid frames."""
	if dim != 1 or not sum or keepdim:
		raise _lib.ConvasrHipError('entropy: only the dim=1, sum=True form used by train.py:756 / train.py:137 is implemented')
	return ops.entropy(log_probs, lengths, eps)


def unpad(x, lens):
	return [e[..., :l] for e, l in zip(x, lens)]


def compute_capacity(model, scale = 1):
	return sum(map(torch.numel, model.parameters())) / scale


def master_module(model):
	from .parallel import DataParallelEngine
	return model.module if isinstance(model, (DataParallelEngine, torch.nn.parallel.DistributedDataParallel, torch.nn.DataParallel)) else model


def reset_bn_running_stats_(model):
	"""models.py:726-733: every batch norm back to mean 0 / variance 1 / zero batches seen, switched to the cumulative moving average
	(momentum None) and to training mode -- the preparation of the reference's statistics re-estimation pass."""
	for m in model.modules():
		if not isinstance(m, nn.modules.batchnorm._BatchNorm):
			continue
		with torch.no_grad():
			m.running_mean.fill_(0.0)
			m.running_var.fill_(1.0)
			m.num_batches_tracked.fill_(0)
		m.momentum = None
		m.train()
	return model


def weighted_mean_entropy(log_probs, lengths = None, dim = -2, eps = 1e-9, eps_id = -1):
	"""models.py:660-682 (the uncertainty measure evaluate_model logs, train.py:137-139): entropy per frame, averaged with weights
	1 - P(silence token) over the valid frames.  log_probs (B, C, T); dim must be the class axis."""
	if log_probs.ndim != 3 or dim not in (-2, 1):
		raise _lib.ConvasrHipError('weighted_mean_entropy: only (B, C, T) log-probs with dim = -2 (the form train.py:137 uses) are implemented')
	return ops.weighted_mean_entropy(log_probs, lengths, eps, eps_id)


def normalize_signal(signal, dim = -1, eps = 1e-5, denom_multiplier = 1.0):
	"""models.py:684-686.  Inside LogFilterBankFrontend.forward the normalisation is fused into the log-mel kernel; this is the
	standalone function (two launches)."""
	if dim not in (-1, signal.ndim - 1):
		raise _lib.ConvasrHipError('normalize_signal: only the time axis (dim = -1) is implemented')
	shape = signal.shape
	return ops.normalize_signal(signal.reshape(-1, shape[-1]), eps, denom_multiplier).reshape(shape)


# ------------------------------------------------------------------------------------------------ mel filterbank (models.py:522)

def _slaney_hz_to_mel(f):
	f = np.asarray(f, dtype = np.float64)
	lin = f * 3.0 / 200.0
	log_region = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4))
	return np.where(f >= 1000.0, log_region, lin)


def _slaney_mel_to_hz(m):
	m = np.asarray(m, dtype = np.float64)
	return np.where(m >= 15.0, 1000.0 * np.exp((m - 15.0) * (np.log(6.4) / 27.0)), m * 200.0 / 3.0)


def slaney_mel_filterbank(sample_rate, nfft, n_mels, fmin = 0.0, fmax = None):
	"""What `librosa.filters.mel(sample_rate, nfft, n_mels=..., fmin=0, fmax=sr/2)` (models.py:522; librosa < 0.10 defaults:
	Slaney scale, area normalisation) evaluates to: (n_mels, nfft//2+1) float32."""
	fmax = sample_rate / 2.0 if fmax is None else float(fmax)
	bins = np.linspace(0.0, sample_rate / 2.0, nfft // 2 + 1)
	edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
	lo, mid, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
	rising = (bins[None, :] - lo) / (mid - lo)
	falling = (hi - bins[None, :]) / (hi - mid)
	tri = np.clip(np.minimum(rising, falling), 0.0, None)
	return (tri * (2.0 / (hi - lo))).astype(np.float32)


# ------------------------------------------------------------------------------------------------ frontend

class LogFilterBankFrontend(nn.Module):
	"""models.py:486-603.  forward(signal (B, T) float or int16, mask = None) -> (B, out_channels, 1 + T // hop) fp32 log-mel."""

	def __init__(self, out_channels, sample_rate, window_size, window_stride, window, dither = 1e-5, dither0 = 0.0, preemphasis = 0.97, eps = FP16_TINY, normalize_signal = True, debug_short_long_records_normalize_signal_multiplier = 1.0, stft_mode = None, window_periodic = True, normalize_features = False, **kwargs):
		super().__init__()
		if stft_mode not in (None, '', 'conv'):
			raise ValueError(f'stft_mode {stft_mode!r}')
		self.debug_short_long_records_normalize_signal_multiplier = float(debug_short_long_records_normalize_signal_multiplier)
		self.stft_mode, self.dither, self.dither0 = stft_mode or None, dither, dither0
		self.preemphasis, self.normalize_signal, self.sample_rate = preemphasis, normalize_signal, sample_rate
		self.win_length = int(window_size * sample_rate)
		self.hop_length = int(window_stride * sample_rate)
		self.nfft = 2 ** math.ceil(math.log2(self.win_length))
		self.freq_cutoff = self.nfft // 2 + 1
		self.register_buffer('window', getattr(torch, window)(self.win_length, periodic = window_periodic).float())
		basis = torch.as_tensor(slaney_mel_filterbank(sample_rate, self.nfft, out_channels, 0.0, int(sample_rate / 2)))
		self.mel = nn.Conv1d(basis.shape[1], basis.shape[0], 1).requires_grad_(False)  # parameter container only
		with torch.no_grad():
			self.mel.weight.copy_(basis.unsqueeze(-1))
			self.mel.bias.fill_(eps)
		# stft_mode = 'conv' (models.py:548-561): the reference evaluates the same STFT as a strided Conv1d with the windowed DFT basis (an
		# ONNX-friendly formulation).  Here both modes run the fused FFT kernel -- the two are the same linear map -- and the basis is kept
		# only as a parameter container, so that a checkpoint saved with the conv formulation ('frontend.stft.weight') loads unchanged.
		self.stft = None
		if self.stft_mode == 'conv':
			basis_ri = torch.view_as_real(torch.fft.fft(torch.eye(self.nfft), dim = 1))[:self.freq_cutoff].permute(2, 0, 1).reshape(-1, 1, self.nfft)
			left = (self.nfft - self.win_length) // 2
			centred = torch.nn.functional.pad(self.window, (left, self.nfft - self.win_length - left))  # librosa.util.pad_center
			self.stft = nn.Conv1d(1, basis_ri.shape[0], self.nfft, bias = False, stride = self.hop_length).requires_grad_(False)
			with torch.no_grad():
				self.stft.weight.copy_(basis_ri * centred)

	def forward(self, signal, mask = None, xlen = None, **kwargs):
		assert signal.ndim == 2
		_lib.require_cuda(signal)
		if xlen is None and mask is not None:
			# a temporal_mask() prefix mask: recover the lengths exactly ((n - 0.5) / T -> ceil -> n)
			n = mask.reshape(mask.shape[0], -1).sum(dim = -1).to(torch.float32)
			xlen = (n - 0.5) / signal.shape[-1]
		return ops.logmel(signal, xlen, self.window, self.mel.weight.view(self.mel.weight.shape[0], -1), self.mel.bias, self.nfft, self.hop_length, preemphasis = self.preemphasis, normalize = self.normalize_signal, denom_multiplier = self.debug_short_long_records_normalize_signal_multiplier)

	@staticmethod
	def compute_output_shape(time_dim_length, kernel_size, stride, padding, dilation = 1):
		return int(math.floor((time_dim_length + 2 * padding - dilation * (kernel_size - 1) - 1) / stride + 1))


class MaskedInstanceNorm1d(nn.InstanceNorm1d):
	"""models.py:688-719: per-(utterance, channel) masked mean / biased std over time, no affine; running statistics only in the un-masked
	nn.InstanceNorm1d form (models.py:711)."""

	def __init__(self, *args, temporal_mask = False, legacy = True, **kwargs):
		super().__init__(*args, **kwargs)
		self.temporal_mask, self.legacy = temporal_mask, legacy
		# legacy = False hands the unmasked case to nn.InstanceNorm1d.forward (models.py:711): (x - mean) / sqrt(biased var + eps), the same
		# expression the legacy branch spells out (models.py:704-710) -- one kernel serves both
		if self.affine:
			raise _lib.ConvasrHipError('MaskedInstanceNorm1d: affine = True is not implemented (no configuration of the reference passes it)')

	def forward(self, x, mask = None, xlen = None, out_dtype = None, pad_time_to = 1):
		_lib.require_cuda(x)
		if x.requires_grad:
			raise _lib.ConvasrHipError('MaskedInstanceNorm1d backward is not implemented (features never require grad on this path)')
		if not self.temporal_mask:
			xlen, mask = None, None
		if xlen is None and mask is not None:
			n = mask.reshape(mask.shape[0], -1).sum(dim = -1).to(torch.float32)
			xlen = (n - 0.5) / x.shape[-1]
		if self.track_running_stats:
			# models.py:711 (legacy = False, no mask: JasperNetSmallTrainableInstanceNorm); the masked branch asserts there are none (713), the legacy one too (698)
			assert xlen is None and not self.legacy, 'running statistics only in the un-masked nn.InstanceNorm1d form (models.py:698, 713)'
			# (torch's _InstanceNorm.forward passes momentum None on as 0.0 and never touches num_batches_tracked)
			return ops.instnorm_running(x, self.running_mean, self.running_var, None, self.momentum or 0.0, self.training, self.eps, out_dtype = out_dtype or x.dtype, pad_time_to = pad_time_to)
		return ops.instnorm(x, xlen, self.eps, out_dtype = out_dtype or x.dtype, pad_time_to = pad_time_to)


# ------------------------------------------------------------------------------------------------ conv block

class ConvSamePadding(nn.Sequential):
	"""models.py:47-77: parameter container; padding = dilation * kernel_size // 2."""

	def __init__(self, in_channels, out_channels, kernel_size, stride, dilation, bias, groups, separable):
		padding = dilation * kernel_size // 2
		if separable:
			# models.py:50-64: grouped conv (with its default bias) -> ReLU -> 1x1 conv; index 0 runs in csrc/grouped.hip, index 2 in the MFMA kernels
			assert dilation == 1
			super().__init__(nn.Conv1d(in_channels, out_channels, kernel_size = kernel_size, stride = stride, padding = padding, dilation = dilation, groups = groups), nn.ReLU(inplace = True), nn.Conv1d(out_channels, out_channels, kernel_size = 1, bias = bias))
		elif groups != 1:
			raise _lib.ConvasrHipError('grouped convolutions outside the separable block (models.py:50-64) are not implemented')
		else:
			super().__init__(nn.Conv1d(in_channels, out_channels, kernel_size = kernel_size, stride = stride, padding = padding, dilation = dilation, groups = groups, bias = bias))

	@property
	def separable(self):
		return len(self) == 3


def _spec_of(conv):
	return ConvSpec(conv.kernel_size[0], conv.stride[0], conv.dilation[0], conv.padding[0])


class ResidualActivation(nn.Module):
	"""models.py:350-371: carries the nonlinearity / dropout configuration (the arithmetic is fused into bn_act kernels)."""

	def __init__(self, nonlinearity, dropout = 0, invertible = False):
		super().__init__()
		# invertible = True (the *Inplace configs, models.py:357-400): the reference recomputes the activation's input from its output to save
		# memory; the VALUES are act(y + sum residuals) followed by dropout either way, which is what the fused kernels compute -- with
		# 288 GB of HBM there is nothing to save, so the flag only records the configuration
		self.nonlinearity, self.dropout, self.invertible = nonlinearity, dropout, invertible

	def extra_repr(self):
		return f'nonlinearity={self.nonlinearity}, dropout={self.dropout}'


class ConvBn1d(nn.Module):
	"""models.py:80-151."""

	def __init__(self, num_channels, kernel_size, stride = 1, dropout = 0, groups = 1, num_channels_residual: typing.List = [], repeat = 1, dilation = 1, separable = False, temporal_mask = True, nonlinearity = ('relu', ), nonlinearity_reference = True, batch_norm_momentum = 0.1, inplace = False):
		super().__init__()
		# inplace = True: the reference swaps in InplaceBatchNorm1d (models.py:402-433: the same statistics, affine map, parameters and
		# state-dict keys, computed in place) and the invertible activation; same values, see ResidualActivation
		self.conv = nn.ModuleList(ConvSamePadding(num_channels[0] if i == 0 else num_channels[1], num_channels[1], kernel_size = kernel_size, stride = stride, dilation = dilation, separable = separable, bias = False, groups = groups) for i in range(repeat))
		self.bn = nn.ModuleList(nn.BatchNorm1d(num_channels[1], momentum = batch_norm_momentum) for i in range(repeat))
		self.conv_residual = nn.ModuleList(nn.Identity() if c is None else nn.Conv1d(c, num_channels[1], kernel_size = 1) for c in num_channels_residual)
		self.bn_residual = nn.ModuleList(nn.Identity() if c is None else nn.BatchNorm1d(num_channels[1], momentum = batch_norm_momentum) for c in num_channels_residual)
		self.activation = ResidualActivation(nonlinearity, dropout, invertible = inplace)
		self.temporal_mask = temporal_mask
		self.compute_dtype = torch.float32
		self.split_dtype = None  # bf16 / fp16: the convs of an fp32 network run as split-operand MFMA convs (JasperNet.set_compute_dtype('bf16x3'))
		self.split_hi_bwd = False  # 'bf16x3f' / 'f16x3f': the backward of a split conv as ONE 16-bit product per gradient (hi planes only)
		self.split_inference = False  # the evaluation path too (set_compute_dtype('bf16x3', inference = True)); off by default: evaluation then runs the exact-fp32 kernels
		self.tapped_output = False  # set by the network when later blocks take this block's output as a residual input: its gradient then has an accumulator the tapping blocks write into (functional.ConvBnActFunction, GRAD_ACC)
		self.single_consumer_output = False  # set by the network when this block's output feeds exactly one conv (no residual taps): enables cross-layer backward fusion
		self.feeds_block = None  # set by the network: weakref to the ConvBn1d whose first conv is the ONLY reader of this block's output (None: the decoder, several readers, or unknown)

	def _planes_out(self, i, last):
		"""Does repeat i hand its output on as split-operand planes only (functional.ConvBnActFunction, `planes_out`)?  Yes when the network runs
		split convs, a gradient is wanted, and the ONE reader of that output is a split-eligible conv: the next repeat of this block, or the first
		conv of the block the network wired behind this one (feeds_block)."""
		if self.split_dtype is None or not (self.training and torch.is_grad_enabled()) or os.environ.get('CONVASR_NO_PLANES_OUT') == '1':
			return False
		if not last:
			nxt = self.conv[i + 1]
		else:
			blk = self.feeds_block() if (self.feeds_block is not None and self.single_consumer_output) else None
			if blk is None or not blk.training or blk.split_dtype != self.split_dtype or not isinstance(blk.bn[0], nn.BatchNorm1d) or not blk.bn[0].training:
				return False
			nxt = blk.conv[0]
		if nxt.separable or not isinstance(nxt[-1], nn.Conv1d):
			return False
		c = nxt[-1]
		return Fn.split_applies(self.split_dtype, self.compute_dtype, _spec_of(c), c.in_channels, c.out_channels) and c.in_channels == self.conv[i][-1].out_channels

	def _cfg(self, i, last):
		conv, bn = self.conv[i][-1], self.bn[i]
		return dict(planes_out = self._planes_out(i, last), spec = _spec_of(conv), bn = bn, res_bn = list(self.bn_residual) if last else [], act = ops.act_args(self.activation.nonlinearity), dropout_p = float(self.activation.dropout) if self.training else 0.0, temporal_mask = self.temporal_mask, compute_dtype = self.compute_dtype, split = self.split_dtype if (self.training and torch.is_grad_enabled()) else None, split_hi_bwd = self.split_hi_bwd, split_eval = self.split_dtype if self.split_inference else None, fuse_bwd = self.training and torch.is_grad_enabled() and (not last or self.single_consumer_output), tappable = last and self.tapped_output and torch.is_grad_enabled())

	def forward(self, x, lengths_fraction = None, residual: typing.List = []):
		_lib.require_cuda(x)
		n = len(self.conv)
		for i in range(n):
			last = i == n - 1
			res = list(residual) if last else []
			if last:
				assert len(self.conv_residual) == len(self.bn_residual) == len(residual)
			cfg = self._cfg(i, last)
			conv, bn = self.conv[i][-1], self.bn[i]
			if self.conv[i].separable:  # grouped conv + bias + ReLU first; what follows is this repeat with its 1x1 conv
				g = self.conv[i][0]
				x = Fn.GroupedConvReluFunction.apply(dict(spec = _spec_of(g), groups = g.groups, compute_dtype = self.compute_dtype), x, g.weight, g.bias)
			bn_live = isinstance(bn, nn.BatchNorm1d)
			if bn_live and bn.training:
				flat = []
				for rc, rbn, rx in zip(self.conv_residual, self.bn_residual, res):
					if isinstance(rc, nn.Identity):
						flat += [rx, None, None, None, None]
					elif not (isinstance(rbn, nn.BatchNorm1d) and rbn.training):
						raise _lib.ConvasrHipError('mixed train/eval batch norms inside one ConvBn1d are not supported')
					else:
						flat += [rx, rc.weight, rc.bias, rbn.weight, rbn.bias]
				x = Fn.ConvBnActFunction.apply(cfg, x, conv.weight, bn.weight, bn.bias, lengths_fraction, *flat)
			else:
				res_live = lambda rc, rbn, rx: rx.requires_grad or (not isinstance(rc, nn.Identity) and any(p.requires_grad for p in list(rc.parameters()) + list(rbn.parameters())))
				wants_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in list(conv.parameters()) + list(bn.parameters())) or any(res_live(rc, rbn, rx) for rc, rbn, rx in zip(self.conv_residual, self.bn_residual, res)))
				if wants_grad:
					# batch norms on their running statistics (bn.eval(): JasperNet.freeze with a gradient still flowing through, or statistics frozen
					# for fine-tuning) under autograd: backward through the per-channel affine map (functional.ConvBnActFrozenStatsFunction)
					flat = []
					for rc, rbn, rx in zip(self.conv_residual, self.bn_residual, res):
						if isinstance(rc, nn.Identity):
							flat += [rx, None, None, None, None]
						elif isinstance(rbn, nn.BatchNorm1d) and rbn.training:
							raise _lib.ConvasrHipError('mixed train/eval batch norms inside one ConvBn1d are not supported')
						else:
							live_bn = isinstance(rbn, nn.BatchNorm1d)
							flat += [rx, rc.weight, rc.bias, rbn.weight if live_bn else None, rbn.bias if live_bn else None]
					fcfg = dict(cfg, bn = bn if bn_live else None, res_bn = list(self.bn_residual) if last else [])
					x = Fn.ConvBnActFrozenStatsFunction.apply(fcfg, x, conv.weight, conv.bias, bn.weight if bn_live else None, bn.bias if bn_live else None, lengths_fraction, *flat)
					continue
				ss = ops.bn_eval_scale_shift(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps) if bn_live else None
				res_list = []
				for rc, rbn, rx in zip(self.conv_residual, self.bn_residual, res):
					if isinstance(rc, nn.Identity):
						res_list.append((rx, None, None, None))
					else:
						rss = ops.bn_eval_scale_shift(rbn.weight, rbn.bias, rbn.running_mean, rbn.running_var, rbn.eps) if isinstance(rbn, nn.BatchNorm1d) else None
						res_list.append((rx, rc.weight, rc.bias, rss))
				x = Fn.ConvBnActEvalFunction.apply(cfg, x, conv.weight, conv.bias, ss, lengths_fraction, res_list)
		return x

	def fuse_conv_bn_eval(self):
		"""models.py:141-151: fold BN running statistics into the conv weights / bias (inference): every (conv, batch norm) pair of the block --
		the repeats' last convs and the 1x1 residual branches -- becomes one conv with a bias, the batch norm an nn.Identity (same module
		slots, so state-dict keys of what remains are the reference's)."""
		fold = nn.utils.fusion.fuse_conv_bn_eval
		for r, (rc, rbn) in enumerate(zip(self.conv_residual, self.bn_residual)):
			if isinstance(rc, nn.Identity) or isinstance(rbn, nn.Identity):
				continue  # an identity residual, or a branch folded earlier
			self.conv_residual[r], self.bn_residual[r] = fold(rc, rbn), nn.Identity()
		for r, (seq, bn) in enumerate(zip(self.conv, self.bn)):
			seq[-1], self.bn[r] = fold(seq[-1], bn), nn.Identity()
		Fn.invalidate_pack_cache()


class Decoder(nn.Sequential):
	"""models.py:23-44: 1x1 conv head (with bias) -> tuple of logits; 'bpe' adds a two-block ConvBn1d head."""

	def __init__(self, input_size, num_classes, type = None):
		if type is None:
			super().__init__(nn.Conv1d(input_size, num_classes[0], kernel_size = 1))
		elif type == 'bpe':
			super().__init__(nn.Conv1d(input_size, num_classes[0], kernel_size = 1), nn.Sequential(ConvBn1d(num_channels = (input_size, input_size), kernel_size = 15), ConvBn1d(num_channels = (input_size, num_classes[1]), kernel_size = 15)))
		else:
			raise ValueError(type)
		self.type = type
		self.compute_dtype = torch.float32
		self.split_dtype = None  # (JasperNet.set_compute_dtype('bf16x3'): the head as a split-operand conv too)
		self.split_hi_bwd = False  # ('bf16x3f': ... with a one-product backward)

	def _head(self, x):
		conv = self[0]
		cfg = dict(spec = _spec_of(conv), compute_dtype = self.compute_dtype, out_dtype = torch.float32, split = self.split_dtype if (self.training and torch.is_grad_enabled()) else None, split_hi_bwd = self.split_hi_bwd)
		return Fn.ConvBiasFunction.apply(cfg, x, conv.weight, conv.bias)

	def forward(self, x):
		if self.type is None:
			return (self._head(x), )
		y2 = x
		for blk in self[1]:
			y2 = blk(y2)
		return self._head(x), ops.convert(y2, torch.float32, True) if y2.dtype != torch.float32 else y2


# ------------------------------------------------------------------------------------------------ encoder

class JasperNet(nn.Module):
	"""models.py:158-347."""

	def __init__(self, num_input_features, num_classes, repeat = 3, num_subblocks = 1, dilation = 1, residual = 'dense', kernel_sizes = [11, 13, 17, 21, 25], kernel_size_prologue = 11, kernel_size_epilogue = 29, base_width = 128, out_width_factors = [2, 3, 4, 5, 6], out_width_factors_large = [7, 8], separable = False, groups = 1, dropout = 0, dropout_prologue = 0.2, dropout_epilogue = 0.4, dropouts = [0.2, 0.2, 0.2, 0.3, 0.3], temporal_mask = True, nonlinearity = ('relu', ), inplace = False, stride1 = 2, stride2 = 1, decoder_type = None, dict = dict, frontend = None, bpe_only = False, normalize_features = True, normalize_features_eps = FP16_TINY, normalize_features_track_running_stats = False, normalize_features_legacy = True, normalize_features_temporal_mask = True, check_time_dim_padded = True, compute_dtype = torch.float32):
		super().__init__()
		self.init_params = {name: repr(value) for name, value in locals().items() if name not in ('self', '__class__')}
		if dropout == 0:
			dropout_prologue, dropout_epilogue, dropouts = 0, 0, [0] * len(dropouts)
		common = {'temporal_mask': temporal_mask, 'nonlinearity': nonlinearity, 'inplace': inplace}  # (the reference's constructor argument named dict shadows the builtin here)
		width = lambda f: f * base_width

		blocks = [ConvBn1d(num_channels = (num_input_features, width(out_width_factors[0])), kernel_size = kernel_size_prologue, dropout = dropout_prologue, stride = stride1, **common)]
		f_in = out_width_factors[0]
		res_channels = []
		for k, p_drop, f_out in zip(kernel_sizes, dropouts, out_width_factors):
			for s in range(num_subblocks):
				c_in, c_out = width(f_in), width(f_out if s == num_subblocks - 1 else f_in)
				if residual == 'dense':
					res_channels = res_channels + [c_in]
				elif residual == 'flat':
					res_channels = [None]
				elif residual:
					res_channels = [c_in]
				else:
					res_channels = []
				blocks.append(ConvBn1d(num_channels = (c_in, c_out), kernel_size = k, dropout = p_drop, repeat = repeat, separable = separable, groups = groups, num_channels_residual = list(res_channels), **common))
			f_in = f_out
		blocks.append(ConvBn1d(num_channels = (width(f_in), width(out_width_factors_large[0])), kernel_size = kernel_size_epilogue, dropout = dropout_epilogue, dilation = dilation, **common))
		blocks.append(ConvBn1d(num_channels = (width(out_width_factors_large[0]), width(out_width_factors_large[1])), kernel_size = 1, dropout = dropout_epilogue, **common))
		self.backbone = nn.ModuleList(blocks)
		self.num_epilogue_modules = 2
		self.frontend = frontend
		self.normalize_features = MaskedInstanceNorm1d(num_input_features, affine = False, eps = normalize_features_eps, track_running_stats = normalize_features_track_running_stats, temporal_mask = normalize_features_temporal_mask, legacy = normalize_features_legacy) if normalize_features else None
		self.decoder = Decoder(width(out_width_factors_large[1]), num_classes, type = decoder_type)
		self.residual, self.dict, self.bpe_only, self.check_time_dim_padded = residual, dict, bpe_only, check_time_dim_padded
		# a block output that is not tapped as a residual and feeds one conv only (the next block, or a single decoder head)
		for i, blk in enumerate(self.backbone):
			tapped = bool(residual) and i < len(self.backbone) - self.num_epilogue_modules - 1
			blk.single_consumer_output = (not tapped) and (i < len(self.backbone) - 1 or len(num_classes) == 1)
			blk.tapped_output = tapped
			blk.feeds_block = weakref.ref(self.backbone[i + 1]) if (not tapped and i < len(self.backbone) - 1) else None  # (a weak reference: a plain attribute would register the block twice)
		self.set_compute_dtype(compute_dtype)

	SPLIT_DTYPES = {'bf16x3': torch.bfloat16, 'f16x3': torch.float16, 'bf16x3f': torch.bfloat16, 'f16x3f': torch.float16}  # (...f: split forward, one-product backward)

	def set_compute_dtype(self, dtype, inference = False):
		"""fp32 (exact-fp32 MFMA path: parity runs), or bf16 / fp16 (16-bit storage of activations and compute weights, MFMA with fp32
		accumulation, fp32 master weights: throughput runs; fp16 is what the reference's apex O1-O3 levels compute in and trains under a
		dynamic loss scaler, convasr_amd.train.LossScaler), or 'bf16x3' / 'f16x3': fp32 storage everywhere, the training convs as
		split-operand products on the 16-bit matrix pipe (three MFMAs per product, fp32-class accuracy: csrc/split3.hip) -- the path that
		meets the reference's fp32 results to 1e-4 in the CTC loss at MFMA rate; 'bf16x3f' / 'f16x3f': that forward (the same loss, bit for bit)
		with the backward of every split conv as ONE 16-bit product per gradient (dy rounded once, x_hi, w_hi: gradients of the plain 16-bit
		path's accuracy, as under the reference's apex O2; two thirds of the backward's matrix work gone); evaluation runs the exact-fp32 kernels unless inference = True
		(the folded / eval-mode convs then run as split convs as well: fp32-class logits at a third of the 16-bit rate instead of the fp32 MFMA rate)."""
		split, hi_bwd = None, False
		if isinstance(dtype, str):
			dtype, split, hi_bwd = torch.float32, self.SPLIT_DTYPES[dtype], dtype.endswith('x3f')
		assert dtype in (torch.float32, ) + ops.HALF_DTYPES
		self.compute_dtype, self.split_dtype = dtype, split
		for m in self.modules():
			if isinstance(m, (ConvBn1d, Decoder)):
				m.compute_dtype = dtype
			if isinstance(m, (ConvBn1d, Decoder)):
				m.split_dtype, m.split_hi_bwd = split, hi_bwd
			if isinstance(m, ConvBn1d):
				m.split_inference = bool(inference) and split is not None
		return self

	def forward(self, x, xlen = None, y = None, ylen = None):
		_lib.require_cuda(x)
		if self.frontend is not None:
			assert (not self.check_time_dim_padded) or (x.shape[-1] % (32 / 2) == 0), 'Shape of input signal is not divisible by 16 '
			x = x.squeeze(1)
			x = self.frontend(x, xlen = xlen) if isinstance(self.frontend, LogFilterBankFrontend) else self.frontend(x, mask = temporal_mask(x, compute_output_lengths(x, xlen)) if xlen is not None else None)
		assert (not self.check_time_dim_padded) or (x.shape[-1] % 32 == 0), 'Shape of features after frontend is not divisible by 32'
		assert x.ndim == 3
		if self.normalize_features is not None:
			# an odd number of frames gets one zero frame appended when the prologue conv will run as its stride-2 fold (functional.Fold2),
			# which needs an even-length input: the conv's output is the same -- the frame lies in its zero padding -- exactly when the
			# output length does not change (odd kernel sizes), which Fold2.wants_even_input checks together with the fold's own envelope
			conv0 = self.backbone[0].conv[0][-1]
			pad_even = (self.compute_dtype in ops.HALF_DTYPES or (self.split_dtype is not None and self.training and torch.is_grad_enabled())) and x.shape[-1] % 2 == 1 and Fn.Fold2.wants_even_input(conv0.weight.shape, _spec_of(conv0), x.shape[-1])
			x = self.normalize_features(x, xlen = xlen, out_dtype = self.compute_dtype, pad_time_to = 2 if pad_even else 1)
		else:
			x = ops.as_cl(x, self.compute_dtype)

		residual = []
		n = len(self.backbone)
		for i, block in enumerate(self.backbone):
			x = block(x, residual = residual, lengths_fraction = xlen)
			if i >= n - self.num_epilogue_modules - 1:
				residual = []
			elif self.residual == 'dense':
				residual = residual + [x]
			elif self.residual:
				residual = [x]
			else:
				residual = []

		logits = self.decoder(x)
		if y is not None and ylen is not None and self.training and torch.is_grad_enabled():
			# the backward pass's transposed weight copies, on a side stream under the CTC recursion that follows (functional.prepack_dgrad_weights)
			cached = getattr(self, '_dgrad_weights', None)
			if cached is None or cached[0] != Fn.structure_epoch():  # (the module tree is walked once, not per step: any fuse_conv_bn_eval -- the network's or a single block's -- replaces conv modules and bumps the epoch)
				convs = [c[-1] for blk in self.modules() if isinstance(blk, ConvBn1d) for c in blk.conv] + [c for blk in self.modules() if isinstance(blk, ConvBn1d) for c in blk.conv_residual if isinstance(c, nn.Conv1d)]
				cached = self._dgrad_weights = (Fn.structure_epoch(), [c.weight for c in convs[1:] if c.stride[0] == 1])
			if self.split_dtype is None:  # (a split-operand network packs both operands of a conv in one launch at its forward pass: functional.split_weight)
				Fn.prepack_dgrad_weights(cached[1], self.compute_dtype)
		log_probs = [Fn.LogSoftmaxFunction.apply(l) for l in logits]
		olen = [ops.output_lengths(xlen, l.shape[0], l.shape[-1], l.device) for l in logits]  # compute_output_lengths (models.py:611-614) as one launch
		aux = {}
		if y is not None and ylen is not None:
			loss = [Fn.ctc_loss(lp, y[:, i], olen[i], ylen[:, i], lp.shape[1] - 1, norm = ylen[:, 0]) for i, lp in enumerate(log_probs)]
			heads = loss if not self.bpe_only else loss[1:]
			if not heads:  # bpe_only with a single head: the reference's sum(loss[1:]) over nothing is 0 (models.py:325)
				aux = dict(loss = torch.zeros_like(loss[0]))
			else:
				aux = dict(loss = heads[0] if len(heads) == 1 else sum(heads[1:], heads[0]))  # (Python's sum() would start from 0 + tensor: an ATen launch per step)
		return self.dict(logits = logits, log_probs = log_probs, olen = olen, **aux)

	def freeze(self, backbone = 0, decoder0 = False, frontend = False):
		"""models.py:328-339: the first `backbone` blocks, the first decoder head and / or the frontend stop learning -- their parameters
		take no gradient and their batch norms are pinned to evaluation mode (a later model.train() does not wake them: train() becomes a no-op)."""
		targets = list(self.backbone[:backbone]) if backbone else []
		if decoder0:
			targets.append(self.decoder[0])
		if frontend and self.frontend is not None:
			targets.append(self.frontend)
		for part in targets:
			for bn in part.modules():
				if isinstance(bn, nn.modules.batchnorm._BatchNorm):
					bn.eval()
					bn.train = lambda training: None
			for p in part.parameters():
				p.requires_grad = False

	def fuse_conv_bn_eval(self, K = None):
		self._dgrad_weights = None
		for block in self.backbone[:K]:
			block.fuse_conv_bn_eval()

	def set_temporal_mask_mode(self, enabled):
		for module in self.modules():
			module.temporal_mask = enabled


# ------------------------------------------------------------------------------------------------ configs (models.py:819-1442)

class Wav2Letter(JasperNet):
	"""models.py:819-855: 18 conv layers, hardtanh(0, 20), no residuals, dilated k=29 epilogue."""

	def __init__(self, num_input_features, num_classes, dropout = 0.2, base_width = 128, nonlinearity = ('hardtanh', 0, 20), kernel_size_prologue = 11, kernel_size_epilogue = 29, kernel_sizes = [11, 13, 17, 21, 25], dilation = 2, num_blocks = 6, decoder_type = None, normalize_features = True, frontend = None, **kwargs):
		super().__init__(num_input_features, num_classes, base_width = base_width, dropout = dropout, dropout_prologue = dropout, dropout_epilogue = dropout, dropouts = [dropout] * num_blocks, kernel_size_prologue = kernel_size_prologue, kernel_size_epilogue = kernel_size_epilogue, kernel_sizes = [kernel_size_prologue] * num_blocks, out_width_factors = [2, 3, 4, 5, 6], out_width_factors_large = [7, 8], residual = False, dilation = dilation, nonlinearity = nonlinearity, decoder_type = decoder_type, normalize_features = normalize_features, frontend = frontend, **kwargs)


# The Wav2Letter family (models.py:858-1369): one constructor signature, thirteen named settings of it.  Per name: the defaults that differ from
# Wav2Letter's own (models.py:819-855) and what is passed on to JasperNet.  `large_kernels`: the blocks take `kernel_sizes` as given instead of
# repeating the prologue's kernel size; `widths` / `widths_large`: out_width_factors (a number n stands for [n] * num_blocks) and
# out_width_factors_large.
_RELU_NO_DROPOUT = dict(dropout = 0.0, nonlinearity = ('relu', ))
_WAV2LETTER_FAMILY = dict(
	Wav2LetterResidual = ('models.py:858-894', dict(), dict(residual = True)),
	Wav2LetterResidualNoDilation = ('models.py:897-933', dict(dilation = 1), dict(residual = True)),
	Wav2LetterResidualBig = ('models.py:936-973', dict(), dict(residual = True, num_subblocks = 2)),
	Wav2LetterDense = ('models.py:976-1012', dict(), dict(residual = 'dense')),
	Wav2LetterDenseNoDilation = ('models.py:1015-1051', dict(dilation = 1), dict(residual = 'dense')),
	Wav2LetterDenseNoDilationInplace = ('models.py:1054-1091: leaky-relu and the in-place memory tricks', dict(dilation = 1, nonlinearity = ('leaky_relu', 0.01)), dict(residual = 'dense', inplace = True)),
	Wav2LetterDenseLargeKernels = ('models.py:1094-1130', dict(), dict(residual = 'dense', large_kernels = True)),
	Wav2LetterDenseNoDilationLargeKernels = ('models.py:1133-1169', dict(dilation = 1), dict(residual = 'dense', large_kernels = True)),
	Wav2LetterDenseBig = ('models.py:1172-1209', dict(), dict(residual = 'dense', num_subblocks = 2)),
	Wav2LetterDenseBigLargeKernelsNoDropoutReLu = ('models.py:1212-1249', _RELU_NO_DROPOUT, dict(residual = 'dense', num_subblocks = 2, large_kernels = True)),
	Wav2LetterDenseBigLargeKernelsNoDilationNoDropoutReLu = ('models.py:1252-1289', dict(dilation = 1, **_RELU_NO_DROPOUT), dict(residual = 'dense', num_subblocks = 2, large_kernels = True)),
	Wav2LetterDenseBigLargeKernelsNoDilationNoTemporalMaskNoDropoutReLu = ('models.py:1292-1330', dict(dilation = 1, **_RELU_NO_DROPOUT), dict(residual = 'dense', num_subblocks = 2, large_kernels = True, temporal_mask = False)),
	Wav2LetterFlat = ('models.py:1333-1369: identity residuals, constant width', dict(kernel_size_prologue = 13), dict(residual = 'flat', widths = 6, widths_large = [16, 16])),
)


def _wav2letter_variant(name, doc, defaults, passed):
	passed = dict(passed)
	large_kernels, widths, widths_large = passed.pop('large_kernels', False), passed.pop('widths', [2, 3, 4, 5, 6]), passed.pop('widths_large', [7, 8])
	d = dict(dropout = 0.2, nonlinearity = ('hardtanh', 0, 20), kernel_size_prologue = 11, dilation = 2, num_blocks = 5)
	d.update(defaults)

	def __init__(self, num_input_features, num_classes, dropout = d['dropout'], base_width = 128, nonlinearity = d['nonlinearity'], kernel_size_prologue = d['kernel_size_prologue'], kernel_size_epilogue = 29, kernel_sizes = [11, 13, 17, 21, 25], dilation = d['dilation'], num_blocks = d['num_blocks'], decoder_type = None, normalize_features = True, frontend = None, **kwargs):
		args = dict(base_width = base_width, dropout = dropout, dropout_prologue = dropout, dropout_epilogue = dropout, dropouts = [dropout] * num_blocks, kernel_size_prologue = kernel_size_prologue, kernel_size_epilogue = kernel_size_epilogue,
			kernel_sizes = kernel_sizes if large_kernels else [kernel_size_prologue] * num_blocks, out_width_factors = [widths] * num_blocks if isinstance(widths, int) else widths, out_width_factors_large = widths_large,
			dilation = dilation, nonlinearity = nonlinearity, decoder_type = decoder_type, normalize_features = normalize_features, frontend = frontend)
		args.update(passed)
		args.update(kwargs)  # (this package's own: compute_dtype, check_time_dim_padded, ...)
		JasperNet.__init__(self, num_input_features, num_classes, **args)

	return type(name, (JasperNet, ), dict(__init__ = __init__, __doc__ = doc, __module__ = __name__))


for _name, (_doc, _defaults, _passed) in _WAV2LETTER_FAMILY.items():
	globals()[_name] = _wav2letter_variant(_name, _doc, _defaults, _passed)
del _name, _doc, _defaults, _passed


class JasperNetSmall(JasperNet):
	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 1, temporal_mask = False, **kwargs)


class JasperNetLarge(JasperNet):
	"""models.py:1407-1409: 'Jasper 10x5'."""

	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 2, repeat = 5, temporal_mask = False, **kwargs)


class JasperNetBig(JasperNet):
	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 2, temporal_mask = False, **kwargs)


class JasperNetBigNoStride(JasperNet):
	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 2, stride1 = 1, temporal_mask = False, **kwargs)


class JasperNetBigBpeOnly(JasperNet):
	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 2, temporal_mask = False, bpe_only = True, **kwargs)


class JasperNetResidualBig(JasperNet):
	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 2, temporal_mask = False, residual = True, **kwargs)


class JasperNetSeparable(JasperNet):
	"""models.py:1372-1374."""

	def __init__(self, *args, separable = True, groups = 128, **kwargs):
		super().__init__(*args, separable = separable, groups = groups, **kwargs)


class JasperNetBigInplace(JasperNet):
	"""models.py:1432-1442."""

	def __init__(self, *args, **kwargs):
		inplace = kwargs.pop('inplace', True)
		super().__init__(*args, num_subblocks = 2, temporal_mask = False, inplace = inplace, nonlinearity = ('leaky_relu', 0.01), **kwargs)


class JasperNetSmallInstanceNorm(JasperNet):
	"""models.py:1382-1391: the feature normalisation as nn.InstanceNorm1d's own forward, without the length mask."""

	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 1, temporal_mask = False, normalize_features_legacy = False, normalize_features_temporal_mask = False, **kwargs)


class JasperNetSmallTrainableInstanceNorm(JasperNet):
	"""models.py:1394-1404: the same with running statistics in the feature normalisation (instance statistics while training, the running
	ones in eval mode: ops.instnorm_running)."""

	def __init__(self, *args, **kwargs):
		super().__init__(*args, num_subblocks = 1, temporal_mask = False, normalize_features_legacy = False, normalize_features_track_running_stats = True, normalize_features_temporal_mask = False, **kwargs)


# ------------------------------------------------------------------------------------------------ wrappers (models.py:736-765)

def _amp_dtype_from_env():
	"""What apex's O1-O3 mean here: fp16 (the reference's arithmetic) unless CONVASR_AMP_DTYPE says bf16."""
	import os
	names = {'fp16': torch.float16, 'float16': torch.float16, 'f16': torch.float16, 'bf16': torch.bfloat16, 'bfloat16': torch.bfloat16}
	name = os.environ.get('CONVASR_AMP_DTYPE', 'fp16').strip().lower()
	if name not in names:
		raise _lib.ConvasrHipError(f'CONVASR_AMP_DTYPE={name!r}: expected one of {sorted(names)}')
	return names[name]


AMP_DTYPE = _amp_dtype_from_env()


def data_parallel_and_autocast(model, optimizer = None, data_parallel = True, opt_level = None, compute_dtype = None, loss_scale = None, **kwargs):
	"""models.py:736-752 (`apex.amp.initialize(model, optimizers, opt_level, **kwargs)`).  One process drives one MI355X here, so there
	is no single-process DataParallel: opt_level selects the compute dtype -- None / '' / 'O0' -> fp32; 'O1' / 'O2' / 'O3' -> fp16
	storage + MFMA with fp32 accumulation and fp32 master weights, as under apex (batch norm runs in fp32 in every mode:
	keep_batchnorm_fp32 is accepted and has nothing left to decide) -- and, given an optimizer, attaches apex's loss scaling to it:
	dynamic for O1 / O2 (train.LossScaler; overflowed steps are skipped and the scale halves), static 1.0 for O3, `loss_scale`
	(a number or 'dynamic') overriding either, exactly apex's kwarg.  compute_dtype = torch.bfloat16 (an extension; env
	CONVASR_AMP_DTYPE=bf16 makes it the default) runs the same kernels on bf16 storage, whose fp32 exponent range needs no loss scale;
	compute_dtype = 'bf16x3' / 'f16x3' (/ 'bf16x3f' / 'f16x3f') selects the split-operand path (JasperNet.set_compute_dtype), the fp16 forms with the
	dynamic loss scaler."""
	amp = opt_level not in (None, '', 'O0')
	dtype = compute_dtype or (AMP_DTYPE if amp else torch.float32)
	master_module(model).set_compute_dtype(dtype)
	flat = getattr(optimizer, 'flat', None)
	if flat is not None:
		from .train import LossScaler
		if loss_scale is None:
			loss_scale = 'dynamic' if ((dtype == torch.float16 and opt_level in ('O1', 'O2')) or dtype in ('f16x3', 'f16x3f')) else None  # ('f16x3': fp32 storage, but the output gradients travel as fp16 planes -- the same range problem, the same cure)
		flat.loss_scaler = None if loss_scale in (None, 1, 1.0) else LossScaler(flat.data.device, loss_scale = loss_scale)
	elif dtype == torch.float16 and master_module(model).training and opt_level in ('O1', 'O2'):
		# apex would scale the loss here; without an arena optimizer (convasr_amd.train.SGD / optimizers.NovoGrad / AdamW) there is nothing
		# to attach the scaler to, and unscaled fp16 gradients of this network underflow
		import warnings
		warnings.warn(f'data_parallel_and_autocast(opt_level = {opt_level!r}): fp16 training WITHOUT loss scaling -- pass a convasr_amd arena optimizer '
			'(its .flat carries the LossScaler), or compute_dtype = torch.bfloat16', RuntimeWarning, stacklevel = 2)
	return model, optimizer


def distributed_data_parallel_and_autocast(model, local_rank, optimizer = None, opt_level = None, synchronize_bn = False, **kwargs):
	"""models.py:755-765: one process per GPU; gradients are all-reduced over RCCL by convasr_amd.parallel.DataParallelEngine."""
	from .parallel import DataParallelEngine
	if synchronize_bn and model.training:
		# (the reference passes synchronize_bn on its evaluation path only, train.py:556-561, where SyncBatchNorm normalises with the
		# running statistics exactly like BatchNorm1d: nothing to convert; its training path, train.py:704, does not pass it)
		raise _lib.ConvasrHipError('synchronize_bn in training mode: the reference trains with per-GPU batch-norm statistics (train.py:704); cross-GPU batch statistics are not implemented')
	model, optimizer = data_parallel_and_autocast(model, optimizer, opt_level = opt_level, **kwargs)
	training = model.training
	engine = DataParallelEngine(model, device = torch.device('cuda', local_rank))
	engine.train(training)
	return engine, optimizer
