"""A structural description of a JasperNet-family module, read off attributes the reference's models.py and convasr_amd.models share
(module names are state-dict keys, so they have to agree anyway): used by make_golden_r5.py on the REFERENCE's classes to write
model_zoo.json and by tests/test_host_cpu.py on this package's classes to compare.  Plain data only (JSON)."""
import torch.nn as nn


def _conv(c):
	return dict(cin = c.in_channels, cout = c.out_channels, k = c.kernel_size[0], stride = c.stride[0], dilation = c.dilation[0], padding = c.padding[0], groups = c.groups, bias = c.bias is not None)


def _bn(b):
	return None if isinstance(b, nn.Identity) else dict(features = b.num_features, momentum = b.momentum, eps = b.eps, affine = b.affine, track = b.track_running_stats)


def _block(blk):
	return dict(
		convs = [[(_conv(m) if isinstance(m, nn.Conv1d) else type(m).__name__) for m in seq] for seq in blk.conv],
		bns = [_bn(b) for b in blk.bn],
		conv_residual = [(None if isinstance(c, nn.Identity) else _conv(c)) for c in blk.conv_residual],
		bn_residual = [_bn(b) for b in blk.bn_residual],
		nonlinearity = list(blk.activation.nonlinearity), dropout = float(blk.activation.dropout), invertible = bool(blk.activation.invertible), temporal_mask = bool(blk.temporal_mask))


def describe(model):
	nf = model.normalize_features
	return dict(
		backbone = [_block(b) for b in model.backbone],
		residual = model.residual, bpe_only = bool(model.bpe_only), check_time_dim_padded = bool(model.check_time_dim_padded),
		normalize_features = None if nf is None else dict(features = nf.num_features, eps = nf.eps, affine = nf.affine, track = nf.track_running_stats, temporal_mask = bool(nf.temporal_mask), legacy = bool(nf.legacy)),
		decoder = [(_conv(m) if isinstance(m, nn.Conv1d) else [_block(b) for b in m]) for m in model.decoder],
		state_dict = {k: list(v.shape) for k, v in model.state_dict().items()},
		num_params = sum(p.numel() for p in model.parameters()))


def fill_parameters(model, seed):
	"""The same parameter values in the reference's module and in this package's, without shipping a state dict: every floating-point entry of
	the state dict, in sorted key order, from its own seeded generator (conv / linear weights ~ N(0, 1 / fan_in), batch-norm weights in
	[0.5, 1.5], biases in [-0.3, 0.3], running means 0 and variances 1 as constructed)."""
	import math
	import torch
	sd = model.state_dict()
	with torch.no_grad():
		for i, key in enumerate(sorted(sd)):
			v = sd[key]
			if not v.is_floating_point() or 'running_' in key or key.startswith('frontend.'):
				continue
			g = torch.Generator().manual_seed(seed * 100003 + i)
			if v.ndim >= 2:
				v.copy_(torch.randn(v.shape, generator = g) / math.sqrt(v[0].numel()))
			elif '.bn' in key and key.endswith('weight'):
				v.copy_(torch.rand(v.shape, generator = g) + 0.5)
			else:
				v.copy_(torch.rand(v.shape, generator = g) * 0.6 - 0.3)


def oracle_plan(desc):
	"""The oracle's layer plan (the output format of oracle.convasr_oracle.jasper_plan) read off a description made by describe()."""
	layers = []
	for blk in desc['backbone']:
		first = blk['convs'][0][0]
		layers.append(dict(cin = first['cin'], cout = blk['convs'][0][-1]['cout'], k = first['k'], stride = first['stride'], dilation = first['dilation'], repeat = len(blk['convs']),
			res = [None if r is None else r['cin'] for r in blk['conv_residual']]))
	head = desc['decoder'][0]
	return dict(layers = layers, residual = desc['residual'], temporal_mask = desc['backbone'][0]['temporal_mask'], nonlinearity = tuple(desc['backbone'][0]['nonlinearity']), num_classes = [head['cout']], c_last = head['cin'])
