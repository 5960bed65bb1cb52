"""CPU oracle for the convasr hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product path (convasr_amd/) never imports it and fails loudly when the
HIP extension is missing.

What it is: a functional (state-dict in, tensors out) plain-torch fp32 restatement of
the reference's arithmetic for the one path SURVEY.md section 8 names:

    waveform -> logmel frontend -> masked instance norm -> [conv1d -> batchnorm ->
    (+residual) -> activation -> dropout(p=0) -> temporal mask] x N -> 1x1 decoder ->
    log_softmax -> CTC loss / gradient -> training step (clip + SGD), greedy decode.

The reference (vadimkantorov/convasr, /root/reference, read-only) keeps that arithmetic
inside PyTorch ATen calls; every function below cites the reference file:line it follows.

Pinning: tests/golden/make_golden.py imports the reference itself in the authoring
container (stubbing its three missing imports) and writes input/output vectors under
tests/golden/*.npz; tests/test_oracle_golden.py checks every function here against those
vectors.  The mel filterbank has no in-container pin against real librosa (librosa is not
installed and the reference does not vendor it): "parity unpinned" against librosa itself
for that one matrix.  It is pinned instead (a) against an independent third-party
restatement of librosa.filters.mel -- transformers.audio_utils.mel_filter_bank, slaney /
slaney, fixtures in tests/golden/mel_hf.npz, agreement to float32 rounding at five
geometries -- (b) by its closed-form properties and (c) by being committed as a fixture
that both the reference run and this oracle consumed.

A second, independent restatement of CTC (float64 numpy alpha-beta recursion) lives in
ctc_loss_numpy() and is checked against torch.nn.functional.ctc_loss on CPU.

"Next" rows restated here as well, each pinned by vectors tests/golden/make_golden_next.py generates by importing
the reference: novograd_step (optimizers.py:66-90 + train.py:777), ctc_alignment (ctc.py:7-75), bucketing_schedule and collate
(datasets.py:357-401, 305-332).
"""
import math
import typing

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# mel filterbank: librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2) as called at
# models.py:522 (librosa < 0.10 positional signature: Slaney scale, norm='slaney').
# --------------------------------------------------------------------------------------


def _hz_to_mel_slaney(f):
	f = np.asarray(f, dtype = np.float64)
	f_sp = 200.0 / 3
	mels = f / f_sp
	min_log_hz = 1000.0
	min_log_mel = min_log_hz / f_sp
	logstep = np.log(6.4) / 27.0
	with np.errstate(divide = 'ignore'):
		log_part = min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep
	return np.where(f >= min_log_hz, log_part, mels)


def _mel_to_hz_slaney(m):
	m = np.asarray(m, dtype = np.float64)
	f_sp = 200.0 / 3
	freqs = f_sp * m
	min_log_hz = 1000.0
	min_log_mel = min_log_hz / f_sp
	logstep = np.log(6.4) / 27.0
	return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(sample_rate: int, nfft: int, n_mels: int, fmin: float = 0.0, fmax: typing.Optional[float] = None) -> np.ndarray:
	"""(n_mels, nfft//2+1) float32 area-normalised triangular filters on the Slaney mel scale."""
	fmax = float(sample_rate) / 2 if fmax is None else float(fmax)
	n_bins = 1 + nfft // 2
	fftfreqs = np.linspace(0, float(sample_rate) / 2, n_bins)
	mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
	mel_f = _mel_to_hz_slaney(mel_pts)
	fdiff = np.diff(mel_f)
	ramps = mel_f[:, None] - fftfreqs[None, :]
	weights = np.zeros((n_mels, n_bins), dtype = np.float64)
	for i in range(n_mels):
		lower = -ramps[i] / fdiff[i]
		upper = ramps[i + 2] / fdiff[i + 1]
		weights[i] = np.maximum(0, np.minimum(lower, upper))
	enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
	weights *= enorm[:, None]
	return weights.astype(np.float32)


# --------------------------------------------------------------------------------------
# lengths / masks: models.py:611-619
# --------------------------------------------------------------------------------------


def compute_output_lengths(time_dim: int, lengths_fraction: typing.Optional[torch.Tensor], batch: int = 1) -> torch.Tensor:
	"""ceil(frac * T) as int64; full length when frac is None (models.py:611-614)."""
	if lengths_fraction is None:
		return torch.full((batch, ), time_dim, dtype = torch.long)
	return (lengths_fraction * time_dim).ceil().long()


def temporal_mask(time_dim: int, lengths: torch.Tensor) -> torch.Tensor:
	"""bool (B, T): t < length (models.py:617-619, without the view to (B,1,..,T))."""
	return torch.arange(time_dim, dtype = lengths.dtype).unsqueeze(0) < lengths.unsqueeze(1)


# --------------------------------------------------------------------------------------
# frontend: models.py:565-597 (+ normalize_signal 684-686)
# --------------------------------------------------------------------------------------


def frontend_config(sample_rate = 16000, window_size = 0.02, window_stride = 0.01):
	win_length = int(window_size * sample_rate)
	hop_length = int(window_stride * sample_rate)
	nfft = 2 ** math.ceil(math.log2(win_length))
	return dict(win_length = win_length, hop_length = hop_length, nfft = nfft, freq_cutoff = nfft // 2 + 1)


def logmel_frontend(
	signal: torch.Tensor,
	xlen: typing.Optional[torch.Tensor],
	window: torch.Tensor,
	mel_weight: torch.Tensor,
	mel_bias: torch.Tensor,
	nfft: int,
	hop_length: int,
	preemphasis: float = 0.97,
	normalize_signal: bool = True,
	stage: str = 'logmel',
	denom_multiplier: float = 1.0
) -> torch.Tensor:
	"""(B, T) waveform -> (B, n_mels, F) log-mel features.

	normalize (models.py:684-686), pre-emphasis (572-573), mask (575, mask built at 290),
	left reflect pad / right zero pad by nfft/2 (577-582), stft center=False (590),
	power (591-594), mel 1x1 conv with eps bias then log (595).
	"""
	assert signal.ndim == 2
	B, T = signal.shape
	win_length = window.shape[0]
	signal = signal if signal.is_floating_point() else signal.to(torch.float32)
	if normalize_signal and signal.numel() > 0:
		signal = signal / ((signal.abs().max(dim = -1, keepdim = True).values + 1e-5) * denom_multiplier)  # models.py:570, 684-686 (debug_short_long_records_normalize_signal_multiplier)
	if preemphasis > 0:
		signal = torch.cat([signal[..., :1], signal[..., 1:] - preemphasis * signal[..., :-1]], dim = -1)
	if xlen is not None:
		mask = temporal_mask(T, compute_output_lengths(T, xlen))
		signal = signal * mask
	pad = nfft // 2
	padded = F.pad(signal.unsqueeze(1), (pad, 0), mode = 'reflect' if pad < T else 'constant').squeeze(1)
	padded = F.pad(padded, (0, pad), mode = 'constant', value = 0)
	if stage == 'padded':
		return padded
	spec = torch.stft(padded, nfft, hop_length = hop_length, win_length = win_length, window = window, center = False, return_complex = True)
	spec = torch.view_as_real(spec)
	power = (spec * spec).sum(dim = -1)
	if stage == 'power':
		return power
	return F.conv1d(power, mel_weight, mel_bias).log()


# --------------------------------------------------------------------------------------
# feature normalisation: models.py:694-719
# --------------------------------------------------------------------------------------


def masked_instance_norm(x: torch.Tensor, mask: typing.Optional[torch.Tensor], eps: float = float(torch.finfo(torch.float16).tiny)) -> torch.Tensor:
	"""(B, C, T) -> same.  mask: bool (B, T) or None (legacy biased-std branch, 704-710)."""
	if mask is None:
		mean = x.mean(dim = -1, keepdim = True)
		xm = x - mean
		std = (xm * xm).mean(dim = -1, keepdim = True).add(eps).sqrt()
		return xm / std
	mask = mask.unsqueeze(1)
	n = mask.int().sum(dim = -1, keepdim = True)
	mean = (x * mask).sum(dim = -1, keepdim = True) / n
	z = mask * (x - mean)
	std = ((z * z).sum(dim = -1, keepdim = True) / n).add(eps).sqrt()
	return z / std


# --------------------------------------------------------------------------------------
# conv block: models.py:47-77 (conv), 111-114 (BN), 127-139 (block), 357-371 (activation)
# --------------------------------------------------------------------------------------


def conv_same_padding(x, weight, bias = None, stride = 1, dilation = 1):
	"""padding = dilation * kernel_size // 2 (models.py:49) -- grows T by 2 for even d*k."""
	K = weight.shape[-1]
	return F.conv1d(x, weight, bias, stride = stride, padding = dilation * K // 2, dilation = dilation)


def split_planes(x, dtype = torch.bfloat16):
	"""The two 16-bit planes the MI355X split-operand path ('bf16x3' / 'f16x3', csrc/split3.hip) carries an fp32 value as:
	hi = rn16(x), lo = rn16(x - hi); x - hi is exact in fp32.  Returned as fp32 tensors holding the rounded values."""
	hi = x.to(dtype).to(torch.float32)
	lo = (x - hi).to(dtype).to(torch.float32)
	return hi, lo


def conv1d_split3(x, weight, bias = None, stride = 1, padding = 0, dilation = 1, dtype = torch.bfloat16, accumulate = torch.float64):
	"""nn.Conv1d (models.py:47-77) computed the way the split-operand path computes it: every product x w as x_hi w_hi + x_lo w_hi + x_hi w_lo
	(x_lo w_lo dropped), each of the three an exact product of two 16-bit values; `accumulate` is the type the sums run in (float64 here
	isolates what the split itself leaves out; the kernels accumulate in fp32).  The restatement of the ARITHMETIC, not of a kernel."""
	xh, xl = (t.to(accumulate) for t in split_planes(x, dtype))
	wh, wl = (t.to(accumulate) for t in split_planes(weight, dtype))
	conv = lambda a, b: F.conv1d(a, b, None, stride = stride, padding = padding, dilation = dilation)
	y = conv(xh, wh) + conv(xl, wh) + conv(xh, wl)
	return y if bias is None else y + bias.to(accumulate).view(1, -1, 1)


def batch_norm(x, gamma, beta, running_mean, running_var, training, momentum = 0.1, eps = 1e-5):
	"""nn.BatchNorm1d semantics; updates running stats in place when training."""
	return F.batch_norm(x, running_mean, running_var, gamma, beta, training, momentum, eps)


def activation(y, nonlinearity):
	"""models.py:368 with dropout p=0: getattr(F, name)(y, *args)."""
	name = nonlinearity[0]
	if name == 'relu':
		return F.relu(y)
	if name == 'hardtanh':
		return F.hardtanh(y, nonlinearity[1], nonlinearity[2])
	if name == 'leaky_relu':
		return F.leaky_relu(y, nonlinearity[1])
	raise ValueError(name)


def jasper_plan(
	num_input_features,
	num_classes,
	repeat = 3,
	num_subblocks = 1,
	dilation = 1,
	residual = 'dense',
	kernel_sizes = (11, 13, 17, 21, 25),
	kernel_size_prologue = 11,
	kernel_size_epilogue = 29,
	base_width = 128,
	out_width_factors = (2, 3, 4, 5, 6),
	out_width_factors_large = (7, 8),
	temporal_mask = True,
	nonlinearity = ('relu', ),
	stride1 = 2,
	**unused
):
	"""Layer plan of JasperNet.__init__ (models.py:201-264) as a list of dicts."""
	plan = []
	in_f = out_width_factors[0]
	plan.append(dict(cin = num_input_features, cout = in_f * base_width, k = kernel_size_prologue, stride = stride1, dilation = 1, repeat = 1, res = []))
	res = []
	for k, out_f in zip(kernel_sizes, out_width_factors):
		for s in range(num_subblocks):
			cin = in_f * base_width
			cout = out_f * base_width if s == num_subblocks - 1 else in_f * base_width
			if residual == 'dense':
				res = res + [cin]
			elif residual == 'flat':
				res = [None]
			elif residual:
				res = [cin]
			else:
				res = []
			plan.append(dict(cin = cin, cout = cout, k = k, stride = 1, dilation = 1, repeat = repeat, res = list(res)))
		in_f = out_f
	plan.append(dict(cin = in_f * base_width, cout = out_width_factors_large[0] * base_width, k = kernel_size_epilogue, stride = 1, dilation = dilation, repeat = 1, res = []))
	plan.append(dict(cin = out_width_factors_large[0] * base_width, cout = out_width_factors_large[1] * base_width, k = 1, stride = 1, dilation = 1, repeat = 1, res = []))
	return dict(layers = plan, residual = residual, temporal_mask = temporal_mask, nonlinearity = tuple(nonlinearity), num_classes = list(num_classes), c_last = out_width_factors_large[1] * base_width)


WAV2LETTER = dict(base_width = 128, kernel_size_prologue = 11, kernel_size_epilogue = 29, kernel_sizes = [11] * 6, out_width_factors = [2, 3, 4, 5, 6], out_width_factors_large = [7, 8], residual = False, dilation = 2, nonlinearity = ('hardtanh', 0, 20))  # models.py:819-855
JASPERNET_LARGE = dict(num_subblocks = 2, repeat = 5, temporal_mask = False)  # models.py:1407-1409
JASPERNET_BIG = dict(num_subblocks = 2, temporal_mask = False)  # models.py:1412-1414
TINY = dict(base_width = 32, kernel_sizes = [11], out_width_factors = [2], out_width_factors_large = [2, 2], residual = False, repeat = 1)


class _StoreAs(torch.autograd.Function):
	"""A tensor that the MI355X path keeps in a reduced-precision type between kernels: rounded to `dtype` on the way forward, and
	its gradient rounded to `dtype` on the way back (the backward kernels store dy / dz in the same type)."""

	@staticmethod
	def forward(ctx, x, dtype):
		ctx.dtype = dtype
		return x.to(dtype).to(x.dtype)

	@staticmethod
	def backward(ctx, g):
		return g.to(ctx.dtype).to(g.dtype), None


class _GradStoreAs(torch.autograd.Function):
	"""Identity forward; the gradient is rounded to `dtype` (fp32 logits whose gradient enters the bf16 decoder dgrad / wgrad)."""

	@staticmethod
	def forward(ctx, x, dtype):
		ctx.dtype = dtype
		return x.view_as(x)

	@staticmethod
	def backward(ctx, g):
		return g.to(ctx.dtype).to(g.dtype), None


def _stored(x, dtype):
	return x if dtype is None else _StoreAs.apply(x, dtype)


def _stored_weight(w, dtype):
	"""Packed compute copy of an fp32 master weight: rounded forward, the gradient stays fp32 (wgrad writes fp32)."""
	return w if dtype is None else w + (w.detach().to(dtype).to(w.dtype) - w.detach())


def _conv_bn_stored(x, w, b, gamma, beta, running_mean, running_var, training, momentum, eps, storage, **conv_args):
	"""conv -> batch norm as the MI355X path rounds it when activations are stored in `storage` (bf16): fp32 accumulation from
	rounded operands, batch statistics from the fp32 accumulators (conv epilogue), the conv output itself stored rounded and
	normalised from that rounded copy.  storage None: plain F.conv1d + F.batch_norm."""
	y = F.conv1d(x, _stored_weight(w, storage), b, **conv_args)
	if storage is None or not training:
		return F.batch_norm(_stored(y, storage), running_mean, running_var, gamma, beta, training, momentum, eps)
	if y.requires_grad:
		y = _GradStoreAs.apply(y, storage)  # the BN-backward kernel writes dy (direct term + both statistics terms, summed in fp32) in the storage type
	n = y.numel() // y.shape[1]
	with torch.no_grad():
		mean_v = y.mean(dim = (0, 2))
		var_v = y.var(dim = (0, 2), unbiased = False)
		running_mean.mul_(1 - momentum).add_(mean_v, alpha = momentum)
		running_var.mul_(1 - momentum).add_(var_v * (n / max(n - 1, 1)), alpha = momentum)
	yq = y + (y.detach().to(storage).to(y.dtype) - y.detach())  # stored rounded; gradient straight through to y, rounded once above
	# VALUES of the statistics: the fp32 accumulators' (conv epilogue).  Their GRADIENT paths run through the stored copy, as in the
	# backward kernels, which form xhat and both reduction terms from the stored y: dy = gamma invstd (g - mean(g) - xhat_q mean(g xhat_q))
	mean_q = yq.mean(dim = (0, 2))
	var_q = yq.var(dim = (0, 2), unbiased = False)
	mean = mean_q + (mean_v - mean_q).detach()
	var = var_q + (var_v - var_q).detach()
	return (yq - mean[None, :, None]) * (var[None, :, None] + eps).rsqrt() * gamma[None, :, None] + beta[None, :, None]


def conv_block(x, sd, prefix, layer, xlen, residual, nonlinearity, use_temporal_mask, training, bn_momentum = 0.1, storage = None):
	"""ConvBn1d.forward (models.py:127-139) over state-dict entries '{prefix}.conv.{j}.0.weight' etc."""
	rep = layer['repeat']
	for j in range(rep):
		res_in = []
		if j == rep - 1:
			assert len(layer['res']) == len(residual)
			for r, (cin_r, xr) in enumerate(zip(layer['res'], residual)):
				if cin_r is None:
					res_in.append(xr)
				else:
					p = f'{prefix}.bn_residual.{r}'
					if p + '.weight' not in sd:  # fused residual conv
						res_in.append(_stored(F.conv1d(xr, _stored_weight(sd[f'{prefix}.conv_residual.{r}.weight'], storage), sd[f'{prefix}.conv_residual.{r}.bias']), storage))
						continue
					res_in.append(_conv_bn_stored(xr, sd[f'{prefix}.conv_residual.{r}.weight'], sd[f'{prefix}.conv_residual.{r}.bias'], sd[p + '.weight'], sd[p + '.bias'], sd[p + '.running_mean'], sd[p + '.running_var'], training, bn_momentum, 1e-5, storage))
		w = sd[f'{prefix}.conv.{j}.0.weight']
		conv_args = dict(stride = layer['stride'], padding = layer['dilation'] * w.shape[-1] // 2, dilation = layer['dilation'])  # models.py:49
		p = f'{prefix}.bn.{j}'
		if p + '.weight' in sd:
			y = _conv_bn_stored(x, w, sd.get(f'{prefix}.conv.{j}.0.bias'), sd[p + '.weight'], sd[p + '.bias'], sd[p + '.running_mean'], sd[p + '.running_var'], training, bn_momentum, 1e-5, storage, **conv_args)
		else:  # conv already fused with its batch norm (fuse_conv_bn_eval): bias, activation and mask run in the conv epilogue, y is never stored
			y = F.conv1d(x, _stored_weight(w, storage), sd.get(f'{prefix}.conv.{j}.0.bias'), **conv_args)
		for r in res_in:
			y = y + r
		x = activation(y, nonlinearity)
		if use_temporal_mask and xlen is not None:
			lengths = compute_output_lengths(x.shape[-1], xlen)
			x = x * temporal_mask(x.shape[-1], lengths).unsqueeze(1)
		x = _stored(x, storage)
	return x


def fuse_conv_bn_eval(sd, eps = 1e-5):
	"""JasperNet.fuse_conv_bn_eval (models.py:141-151, 341-343 -> torch.nn.utils.fusion.fuse_conv_bn_eval): fold every batch norm's
	running statistics and affine into the conv before it; returns a new state dict without the batch-norm entries."""
	import re
	out = {k: v for k, v in sd.items()}
	for k in list(sd):
		m = re.match(r'^(.*)\.conv\.(\d+)\.0\.weight$', k)
		r = re.match(r'^(.*)\.conv_residual\.(\d+)\.weight$', k)
		if m:
			conv_p, bn_p = f'{m.group(1)}.conv.{m.group(2)}.0', f'{m.group(1)}.bn.{m.group(2)}'
		elif r:
			conv_p, bn_p = f'{r.group(1)}.conv_residual.{r.group(2)}', f'{r.group(1)}.bn_residual.{r.group(2)}'
		else:
			continue
		if bn_p + '.running_var' not in sd:
			continue
		w, b = sd[conv_p + '.weight'], sd.get(conv_p + '.bias')
		scale = sd[bn_p + '.weight'] * torch.rsqrt(sd[bn_p + '.running_var'] + eps)
		out[conv_p + '.weight'] = w * scale.reshape(-1, 1, 1)
		out[conv_p + '.bias'] = ((b if b is not None else torch.zeros_like(scale)) - sd[bn_p + '.running_mean']) * scale + sd[bn_p + '.bias']
		for name in ('.weight', '.bias', '.running_mean', '.running_var', '.num_batches_tracked'):
			out.pop(bn_p + name, None)
	return out


def jasper_forward(sd, plan, x, xlen = None, y = None, ylen = None, frontend = None, training = True, normalize_features = True, storage = None, frozen = None, normalize_features_temporal_mask = True, normalize_features_running = False):
	"""JasperNet.forward (models.py:282-326).  sd: state dict (tensors, BN buffers are updated in place when
	training), plan: jasper_plan(...), frontend: dict(window, nfft, hop_length) or None (x is features).
	storage = torch.bfloat16 restates the same algorithm with the MI355X throughput path's storage precision: activations, conv
	outputs, packed weights and the gradients flowing between layers rounded to bf16 where the HIP kernels store them in bf16,
	every sum accumulated in fp32 (see _conv_bn_stored); storage = None is the reference's fp32 arithmetic.
	frozen = dict(backbone = k, decoder0 = bool): JasperNet.freeze (models.py:328-339) -- the batch norms of the first k blocks run on
	their running statistics (module.eval(), models.py:333) while everything else stays in training mode."""
	n_frozen = max((frozen or {}).get('backbone', 0) or 0, (frozen or {}).get('bn_stats', 0) or 0)  # bn_stats = k: the batch norms of the first k blocks in eval mode (`bn.eval()`), their parameters still trainable
	if frontend is not None:
		x = logmel_frontend(x, xlen, sd['frontend.window'], sd['frontend.mel.weight'], sd['frontend.mel.bias'], frontend['nfft'], frontend['hop_length'])
	assert x.ndim == 3
	if normalize_features:
		# (normalize_features_temporal_mask = False: JasperNetSmallInstanceNorm, models.py:1382-1391 -- MaskedInstanceNorm1d ignores the mask, 696)
		mask = temporal_mask(x.shape[-1], compute_output_lengths(x.shape[-1], xlen)) if xlen is not None and normalize_features_temporal_mask else None
		if normalize_features_running:
			# JasperNetSmallTrainableInstanceNorm (models.py:1394-1404): MaskedInstanceNorm1d falls through to nn.InstanceNorm1d.forward (711), i.e.
			# F.instance_norm with the module's running buffers (updated in place while training, used in eval mode), momentum 0.1, no affine
			x = F.instance_norm(x.float(), sd['normalize_features.running_mean'], sd['normalize_features.running_var'], None, None, training, 0.1, float(torch.finfo(torch.float16).tiny))
		else:
			x = masked_instance_norm(x if x.dtype == torch.float64 else x.float(), mask)  # (float64 only in precision experiments)
	x = _stored(x, storage) if x.requires_grad else (x if storage is None else x.to(storage).to(x.dtype))
	residual = []
	L = len(plan['layers'])
	for i, layer in enumerate(plan['layers']):
		x = conv_block(x, sd, f'backbone.{i}', layer, xlen, residual, plan['nonlinearity'], plan['temporal_mask'], training and i >= n_frozen, storage = storage)
		if i >= L - 2 - 1:
			residual = []
		elif plan['residual'] == 'dense':
			residual = residual + [x]
		elif plan['residual']:
			residual = [x]
		else:
			residual = []
	logits = F.conv1d(x, _stored_weight(sd['decoder.0.weight'], storage), sd['decoder.0.bias'])
	if storage is not None and logits.requires_grad:
		logits = _GradStoreAs.apply(logits, storage)
	log_probs = F.log_softmax(logits, dim = 1)
	log_probs = log_probs if log_probs.dtype == torch.float64 else log_probs.float()  # models.py:316 (.to(float32)); float64 only in precision experiments
	olen = compute_output_lengths(logits.shape[-1], xlen.float() if xlen is not None else None, batch = logits.shape[0])
	out = dict(logits = logits, log_probs = log_probs, olen = olen)
	if y is not None and ylen is not None:
		out['loss'] = ctc_loss(log_probs, y[:, 0], olen, ylen[:, 0]) / ylen[:, 0]
	return out


# --------------------------------------------------------------------------------------
# CTC: call site models.py:323; arithmetic = torch.nn.functional.ctc_loss (ATen, third-party)
# --------------------------------------------------------------------------------------


def ctc_loss(log_probs_bct, targets, olen, ylen, blank = None):
	"""F.ctc_loss exactly as models.py:323 calls it: (B,C,t) log-probs permuted to (t,B,C), blank = C-1, reduction none."""
	blank = log_probs_bct.shape[1] - 1 if blank is None else blank
	return F.ctc_loss(log_probs_bct.permute(2, 0, 1), targets, olen, ylen, blank = blank, reduction = 'none')


def ctc_loss_numpy(log_probs_bct: np.ndarray, targets: np.ndarray, olen: np.ndarray, ylen: np.ndarray, blank: typing.Optional[int] = None):
	"""Independent float64 alpha-beta restatement (Graves 2006 eq. 6-16, the algorithm ATen implements).

	Returns (nll (B,), grad (B,C,t)) where grad is d nll / d log_probs as ATen defines it for log-softmax
	inputs: exp(lp) - exp(logsum_{s:l'_s=c}(alpha+beta) + nll - lp) for t < olen, 0 for t >= olen.
	"""
	lp = np.asarray(log_probs_bct, dtype = np.float64)
	B, C, T = lp.shape
	blank = C - 1 if blank is None else blank
	nll = np.zeros(B)
	grad = np.zeros_like(lp)
	NEG = -np.inf

	def lse(*xs):
		m = max(xs)
		if m == NEG:
			return NEG
		return m + math.log(sum(math.exp(x - m) for x in xs))

	for b in range(B):
		Tb, S = int(olen[b]), int(ylen[b])
		ext = [blank] * (2 * S + 1)
		ext[1::2] = [int(c) for c in targets[b, :S]]
		L = len(ext)
		alpha = np.full((Tb, L), NEG)
		beta = np.full((Tb, L), NEG)
		if Tb == 0:
			nll[b] = 0.0 if S == 0 else np.inf
			continue
		alpha[0, 0] = lp[b, blank, 0]
		if L > 1:
			alpha[0, 1] = lp[b, ext[1], 0]
		for t in range(1, Tb):
			for s in range(L):
				a = [alpha[t - 1, s]]
				if s >= 1:
					a.append(alpha[t - 1, s - 1])
				if s >= 2 and ext[s] != blank and ext[s] != ext[s - 2]:
					a.append(alpha[t - 1, s - 2])
				alpha[t, s] = lse(*a) + lp[b, ext[s], t]
		ll = lse(alpha[Tb - 1, L - 1], alpha[Tb - 1, L - 2]) if L > 1 else alpha[Tb - 1, 0]
		nll[b] = -ll
		beta[Tb - 1, L - 1] = lp[b, blank, Tb - 1]
		if L > 1:
			beta[Tb - 1, L - 2] = lp[b, ext[L - 2], Tb - 1]
		for t in range(Tb - 2, -1, -1):
			for s in range(L):
				a = [beta[t + 1, s]]
				if s + 1 < L:
					a.append(beta[t + 1, s + 1])
				if s + 2 < L and ext[s] != blank and ext[s] != ext[s + 2]:
					a.append(beta[t + 1, s + 2])
				beta[t, s] = lse(*a) + lp[b, ext[s], t]
		for t in range(Tb):
			acc = np.full(C, NEG)
			for s in range(L):
				acc[ext[s]] = lse(acc[ext[s]], alpha[t, s] + beta[t, s])
			with np.errstate(invalid = 'ignore'):
				grad[b, :, t] = np.exp(lp[b, :, t]) - np.exp(acc + nll[b] - lp[b, :, t])
	return nll, grad


# --------------------------------------------------------------------------------------
# entropy metric: models.py:645-657; greedy decode: transcript_generators.py:27-93
# --------------------------------------------------------------------------------------


def entropy(log_probs_bct, lengths = None, eps = 1e-9):
	e = -(log_probs_bct.exp() * log_probs_bct).sum(dim = 1)
	if lengths is None:
		return e.mean(dim = -1)
	e = e * temporal_mask(e.shape[-1], lengths)
	return e.sum(dim = -1) / (eps + lengths.type_as(log_probs_bct))


def weighted_mean_entropy(log_probs_bct, lengths = None, eps = 1e-9, eps_id = -1):
	"""models.py:660-682: entropy per frame, averaged over frames with weights 1 - P(silence token), masked by lengths."""
	prob = log_probs_bct.exp()
	e = -(prob * log_probs_bct).sum(dim = 1)
	weights = 1 - prob[:, eps_id]
	if lengths is not None:
		weights = weights * temporal_mask(e.shape[-1], lengths)
	return (e * weights).sum(dim = -1) / (eps + weights.sum(dim = -1))


def normalize_signal(signal, eps = 1e-5, denom_multiplier = 1.0):
	"""models.py:684-686."""
	signal_max = signal.abs().max(dim = -1, keepdim = True).values + eps
	return signal / (signal_max * denom_multiplier) if signal.numel() > 0 else signal


CHAR_LEGACY_ALPHABET = 'абвгдеёжзийклмнопрстуфхцчшщъыьэюя'  # configs/ru_text_config.json:10


def char_legacy_vocab(alphabet = CHAR_LEGACY_ALPHABET):
	"""text_tokenizers.py:8-21: alphabet + ['*', '.', '2', ' ', '|'] -> 38 symbols, space 36, eps/blank 37."""
	return list(alphabet) + ['*', '.', '2', ' ', '|']


def greedy_decode(log_probs_bct, olen = None, vocab = None, blank_amount_to_space = 10):
	"""GreedyCTCGenerator.generate without timestamps (transcript_generators.py:27-93): returns one string per
	utterance (with time_stamps None the generator emits a single segment per utterance)."""
	vocab = vocab or char_legacy_vocab()
	eps_id, space_id = len(vocab) - 1, len(vocab) - 2
	idx = log_probs_bct.argmax(dim = 1).tolist()
	out = []
	for i, sample in enumerate(idx):
		n = int(olen[i]) if olen is not None else len(sample)
		t = 0
		while t < len(sample) and sample[t] in (eps_id, space_id):
			t += 1
		if t >= len(sample):
			out.append('')
			continue
		tokens = [eps_id]
		allow_repeat = False
		count_eps = 0
		for t in range(t, n):
			c = sample[t]
			if c == eps_id and tokens[-1] == space_id:
				continue
			if c == eps_id:
				allow_repeat = True
				count_eps += 1
				if count_eps >= blank_amount_to_space and tokens[-1] != space_id:
					tokens.append(space_id)
				continue
			elif c == tokens[-1] and not allow_repeat:
				continue
			allow_repeat = False
			tokens.append(c)
			count_eps = 0
		out.append(''.join(vocab[c] for c in tokens[1:]))
	return out


# --------------------------------------------------------------------------------------
# training step: train.py:745-783 with SGD(momentum, weight_decay) (train.py:657-662)
# --------------------------------------------------------------------------------------


def ctc_alignment(log_probs_tbc: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor, target_lengths: torch.Tensor, blank: int = 0) -> torch.Tensor:
	"""Forced alignment as ctc.py:7-75 defines it, restated one utterance at a time.

	The forward variable is the SUM recursion (logsumexp over stay / s-1 / s-2, forbidden moves and unreachable states held
	at finfo.min, not -inf: ctc.py:27-30,45-46); the back-pointer of (t, s) is the argmax over those three predecessors
	(first maximum: 0 = stay, 1 = s-1, 2 = s-2, ctc.py:47).  The sweep runs over ALL T frames of the padded batch and the end
	state (last label vs trailing blank) is picked from the column at T-1 (ctc.py:53-58), the walk starts at frame
	input_lengths-1 (ctc.py:58).  Result (B, S_max) int64: for label j the LAST frame the path spends in its state (the
	reference scatters t in increasing order, ctc.py:72-75); 0 for padded labels."""
	T, B, C = log_probs_tbc.shape
	S_max = targets.shape[1]
	zero = torch.tensor(torch.finfo(torch.float16).min if log_probs_tbc.dtype is torch.float16 else torch.finfo(torch.float32).min, dtype = log_probs_tbc.dtype)
	out = torch.zeros(B, S_max, dtype = torch.long)
	for b in range(B):
		S, Tb = int(target_lengths[b]), int(input_lengths[b])
		ext = torch.full((2 * S + 1, ), blank, dtype = torch.long)
		ext[1::2] = targets[b, :S]
		L = len(ext)
		allow2 = torch.zeros(L, dtype = torch.bool)
		allow2[2:] = ext[2:] != ext[:-2]
		alpha = torch.full((L, ), float(zero), dtype = log_probs_tbc.dtype)
		alpha[0] = log_probs_tbc[0, b, blank]
		if L > 1:
			alpha[1] = log_probs_tbc[0, b, ext[1]]
		back = torch.zeros(T, L, dtype = torch.long)
		for t in range(1, T):
			stay = alpha
			one = torch.cat([zero.reshape(1), alpha[:-1]])
			two = torch.where(allow2, torch.cat([zero.reshape(1).expand(2), alpha[:-2]]), zero)
			prev = torch.stack([stay, one, two])
			back[t] = prev.argmax(dim = 0)
			alpha = log_probs_tbc[t, b, ext] + prev.logsumexp(dim = 0)
		if S == 0:
			continue
		s = 2 * S - 1 + int(alpha[2 * S] > alpha[2 * S - 1])  # argmax of (last label, trailing blank): first maximum
		last = {}
		for t in range(Tb - 1, -1, -1):
			last.setdefault(s, t)
			if t > 0:
				s -= int(back[t, s])
		for j in range(S):
			out[b, j] = last.get(2 * j + 1, 0)
	return out


def bucketing_schedule(bucket: torch.Tensor, batch_size: int, world_size: int, epoch: int) -> torch.Tensor:
	"""BucketingBatchSampler.set_epoch (datasets.py:370-393): (num_batches, batch_size) example indices for one epoch.

	Per bucket (ascending id): pad the bucket with randomly repeated members up to a multiple of batch_size * world_size, shuffle,
	cut into batches; then shuffle GROUPS of world_size consecutive batches (so the world_size ranks of one iteration draw from one
	bucket).  All randomness comes from one torch.Generator seeded with the epoch, consumed in exactly this order."""
	rng = torch.Generator()
	rng.manual_seed(epoch)
	unit = batch_size * world_size
	per_bucket = []
	for k in bucket.unique():
		members = (bucket == k).nonzero(as_tuple = True)[0]
		need = int(math.ceil(len(members) / unit)) * unit
		repeats = torch.randint(0, len(members), size = (need - len(members), ), generator = rng)
		padded = torch.cat([members, members[repeats]])
		per_bucket.append(padded[torch.randperm(len(padded), generator = rng)].reshape(-1, batch_size))
	batches = torch.cat(per_bucket)
	group_order = torch.randperm(len(batches) // world_size, generator = rng)
	order = torch.arange(len(batches)).view(-1, world_size)[group_order].flatten()
	return batches[order]


def collate(samples, time_padding_multiple: int = 128, speaker_missing: int = 0):
	"""AudioTextDataset.collate_fn (datasets.py:305-332) for samples (speaker (S,), x (C, T), *targets (L,)):
	returns (s (B, Smax), x (B, C, Tpad), xlen (B,) fractions of Tpad, y (B, n_targets, Lpad), ylen (B, n_targets));
	T and L are padded up to multiples of time_padding_multiple, S to the batch maximum."""
	up = lambda n, m: int(math.ceil(n / m)) * m
	B, n_t = len(samples), len(samples[0]) - 2
	Smax = max(b[0].shape[-1] for b in samples)
	Tpad = up(max(b[1].shape[-1] for b in samples), time_padding_multiple)
	Lpad = max(up(max(b[2 + j].shape[-1] for b in samples), time_padding_multiple) for j in range(n_t)) if n_t else 0
	s = torch.full((B, Smax), speaker_missing, dtype = torch.int64)
	x = torch.zeros(B, len(samples[0][1]), Tpad, dtype = samples[0][1].dtype)
	y = torch.zeros(B, n_t, Lpad, dtype = torch.int64)
	xlen, ylen = torch.zeros(B, dtype = torch.float32), torch.zeros(B, n_t, dtype = torch.int64)
	for k, (sp, sx, *sy) in enumerate(samples):
		xlen[k] = sx.shape[-1] / Tpad if Tpad > 0 else 1.0
		x[k, ..., :sx.shape[-1]] = sx
		s[k, :sp.shape[-1]] = sp
		for j, t in enumerate(sy):
			y[k, j, :t.shape[-1]] = t
			ylen[k, j] = len(t)
	return s, x, xlen, y, ylen


def novograd_step(params, grads, state, lr = 1.0, betas = (0.95, 0.98), eps = 1e-8, weight_decay = 0.0, dampening = False, max_norm = None):
	"""torch.nn.utils.clip_grad_norm_ (train.py:777) followed by NovoGrad.step (optimizers.py:66-90), functional.

	params / grads: lists of tensors (params updated in place, grads left untouched); state: dict that carries 'ema' (one 0-d
	tensor per parameter: the EMA of the squared gradient NORM of that tensor) and 'mom' between calls.  Returns the total
	gradient norm before clipping."""
	total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
	coef = 1.0
	if max_norm is not None:
		coef = float(torch.clamp(max_norm / (total + 1e-6), max = 1.0))
	first = 'ema' not in state
	if first:
		state['ema'], state['mom'] = [None] * len(params), [None] * len(params)
	for i, (p, g) in enumerate(zip(params, grads)):
		g = g * coef
		g2 = (g ** 2).sum()
		state['ema'][i] = g2 if first else state['ema'][i] * betas[1] + g2 * (1.0 - betas[1])
		d = g / (state['ema'][i] + eps).sqrt()
		if weight_decay > 0:
			d = d + weight_decay * p
		if dampening:
			d = d * (1 - betas[0])
		state['mom'][i] = d if first else state['mom'][i] * betas[0] + d
		p.sub_(lr * state['mom'][i])
	return total


def train_step(sd, plan, x, xlen, y, ylen, frontend = None, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3, max_norm = 100.0, momentum_buffers = None, nesterov = False, storage = None, frozen = None):
	"""One iteration of the reference loop with accumulate=1: forward (748), loss = mean(loss * ylen) (755),
	backward (774), clip_grad_norm_ (777), SGD step (780).  Returns dict(loss, loss_cur, entropy, grad_norm, grads);
	sd parameters and momentum_buffers are updated in place.  frozen: see jasper_forward; frozen parameters get no gradient and
	no update (models.py:337-338: requires_grad = False; torch.optim.SGD skips parameters without a gradient, weight decay included)."""
	fz = frozen or {}
	frozen_prefixes = tuple(f'backbone.{i}.' for i in range(fz.get('backbone', 0) or 0)) + (('decoder.0.', ) if fz.get('decoder0') else ())
	names = [k for k, v in sd.items() if v.is_floating_point() and not k.startswith('frontend.') and 'running_' not in k and not k.startswith(frozen_prefixes)]
	for k in names:
		sd[k].requires_grad_(True)
		sd[k].grad = None
	out = jasper_forward(sd, plan, x, xlen, y, ylen, frontend = frontend, training = True, storage = storage, frozen = frozen)
	loss_vec = out['loss']
	loss = (loss_vec * ylen[:, 0]).mean()
	loss_cur = loss_vec.mean()
	ent = entropy(out['log_probs'].detach(), out['olen']).mean()
	res = dict(loss = loss.detach(), loss_cur = loss_cur.detach(), entropy = ent, loss_vec = loss_vec.detach(), logits = out['logits'].detach(), log_probs = out['log_probs'].detach(), olen = out['olen'])
	if not (torch.isinf(loss_cur) or torch.isnan(loss_cur)):
		loss.backward()
		params = [sd[k] for k in names]
		res['grad_norm'] = torch.nn.utils.clip_grad_norm_(params, max_norm).detach()
		res['grads'] = {k: sd[k].grad.detach().clone() for k in names}
		with torch.no_grad():
			for k in names:
				p = sd[k]
				g = p.grad
				if weight_decay != 0:
					g = g.add(p, alpha = weight_decay)
				if momentum != 0:
					if momentum_buffers is not None:
						buf = momentum_buffers.get(k)
						if buf is None:
							buf = momentum_buffers[k] = g.clone()
						else:
							buf.mul_(momentum).add_(g)
						g = g.add(buf, alpha = momentum) if nesterov else buf
				p.add_(g, alpha = -lr)
	for k in names:
		sd[k].requires_grad_(False)
		sd[k].grad = None
	return res


# --------------------------------------------------------------------------------------
# fresh parameters with the reference's state-dict layout -- used by bench.py's
# cpu_baseline leg and by tests that need a full-size model without the reference present.
# --------------------------------------------------------------------------------------


def init_state_dict(plan, seed = 1, frontend = None, num_input_features = 64, sample_rate = 16000):
	g = torch.Generator().manual_seed(seed)
	sd = {}

	def conv_w(cout, cin, k):
		bound = 1.0 / math.sqrt(cin * k)
		return (torch.rand(cout, cin, k, generator = g) * 2 - 1) * bound

	def bn(prefix, c):
		sd[prefix + '.weight'] = torch.ones(c)
		sd[prefix + '.bias'] = torch.zeros(c)
		sd[prefix + '.running_mean'] = torch.zeros(c)
		sd[prefix + '.running_var'] = torch.ones(c)
		sd[prefix + '.num_batches_tracked'] = torch.zeros((), dtype = torch.long)

	for i, layer in enumerate(plan['layers']):
		for j in range(layer['repeat']):
			cin = layer['cin'] if j == 0 else layer['cout']
			sd[f'backbone.{i}.conv.{j}.0.weight'] = conv_w(layer['cout'], cin, layer['k'])
			bn(f'backbone.{i}.bn.{j}', layer['cout'])
		for r, cin_r in enumerate(layer['res']):
			if cin_r is not None:
				sd[f'backbone.{i}.conv_residual.{r}.weight'] = conv_w(layer['cout'], cin_r, 1)
				sd[f'backbone.{i}.conv_residual.{r}.bias'] = (torch.rand(layer['cout'], generator = g) * 2 - 1) / math.sqrt(cin_r)
				bn(f'backbone.{i}.bn_residual.{r}', layer['cout'])
	sd['decoder.0.weight'] = conv_w(plan['num_classes'][0], plan['c_last'], 1)
	sd['decoder.0.bias'] = (torch.rand(plan['num_classes'][0], generator = g) * 2 - 1) / math.sqrt(plan['c_last'])
	if frontend is not None:
		sd['frontend.window'] = torch.hann_window(frontend['win_length'], periodic = True).float()
		sd['frontend.mel.weight'] = torch.as_tensor(mel_filterbank(sample_rate, frontend['nfft'], num_input_features)).unsqueeze(-1)
		sd['frontend.mel.bias'] = torch.full((num_input_features, ), float(torch.finfo(torch.float16).tiny))
	return sd
