"""Host-side logic that needs no GPU: optimizer state in reference shapes, the fp16 / bf16 mapping of the apex opt levels, the
loss scaler's host interface, graded gradient buckets, the stride-2 fold's padding rule, the accepted configuration flags."""
import os

import pytest
import torch

import convasr_amd as ca
from convasr_amd import functional as Fn
from convasr_amd.functional import ConvSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _small():
	torch.manual_seed(0)
	return torch.nn.Sequential(torch.nn.Conv1d(8, 16, 3), torch.nn.BatchNorm1d(16), torch.nn.Conv1d(16, 4, 1))


def test_optimizer_state_dicts_are_per_parameter_in_reference_shapes():
	"""SGD / NovoGrad state dicts (format 2): momentum as one tensor per parameter in the parameter's logical shape -- independent of the
	arena's element order -- and a round trip restores it; a flat arena-ordered buffer (the untagged pre-format-2 layout) is rejected
	with a message instead of being loaded into the wrong elements."""
	m = _small()
	flat = ca.train.FlatParameters(m)
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9)
	opt.momentum_buffer.copy_(torch.arange(flat.numel, dtype = torch.float32))
	sd = opt.state_dict()
	assert sd['format'] == 2 and [tuple(t.shape) for t in sd['momentum_buffer']] == [tuple(p.shape) for p in flat.params]
	opt2 = ca.train.SGD(ca.train.FlatParameters(_small()), lr = 1e-2, momentum = 0.9)
	opt2.load_state_dict(sd)
	for a, b in zip(opt2.flat.param_views(opt2.momentum_buffer), sd['momentum_buffer']):
		assert torch.equal(a, b)
	with pytest.raises(ValueError, match = 'layout tag'):
		opt2.load_state_dict(dict(sd, momentum_buffer = torch.zeros(flat.numel)))
	with pytest.raises(ValueError, match = 'shape'):
		opt2.load_state_dict(dict(sd, momentum_buffer = [t.flatten() for t in sd['momentum_buffer']]))
	ng = ca.optimizers.NovoGrad(ca.train.FlatParameters(_small()), lr = 1e-2)
	ng.momentum_buffer.normal_()
	ng.grads_ema[0, :ng.n_seg] = torch.arange(ng.n_seg, dtype = torch.float32)
	ng.grads_ema[0, ng.n_seg] = 3.0
	nsd = ng.state_dict()
	assert nsd['steps_applied'] == 3 and nsd['grads_ema'].numel() == ng.n_seg
	ng2 = ca.optimizers.NovoGrad(ca.train.FlatParameters(_small()), lr = 1e-2)
	ng2.load_state_dict(nsd)
	assert torch.equal(ng2.grads_ema[0, :ng2.n_seg], ng.grads_ema[0, :ng.n_seg]) and float(ng2.grads_ema[0, ng2.n_seg]) == 3.0
	assert all(torch.equal(a, b) for a, b in zip(ng2.flat.param_views(ng2.momentum_buffer), ng.flat.param_views(ng.momentum_buffer)))  # (the alignment padding between parameters is not state)
	views = ng2.state
	assert all(tuple(views[p]['momentum_buffer'].shape) == tuple(p.shape) for p in ng2.flat.params)
	ng2.load_state_dict(dict(nsd, grads_ema = torch.cat([nsd['grads_ema'], torch.tensor([5.0])])))  # the earlier layout: counter behind the EMAs
	assert float(ng2.grads_ema[0, ng2.n_seg]) == 5.0


def test_opt_levels_select_fp16_like_apex_and_attach_a_loss_scaler():
	model = ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, check_time_dim_padded = False)
	flat = ca.train.FlatParameters(model)
	opt = ca.train.SGD(flat)
	assert ca.train.amp_state_dict(opt) == {}
	for level, dtype, scaler in ((None, torch.float32, False), ('O0', torch.float32, False), ('O1', torch.float16, True), ('O2', torch.float16, True), ('O3', torch.float16, False)):
		ca.models.data_parallel_and_autocast(model, opt, opt_level = level)
		assert model.compute_dtype == dtype and (flat.loss_scaler is not None) == scaler, level
	ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2', compute_dtype = torch.bfloat16)
	assert model.compute_dtype == torch.bfloat16 and flat.loss_scaler is None
	ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2', loss_scale = 128.0, keep_batchnorm_fp32 = True)
	assert flat.loss_scaler.state_dict() == dict(loss_scale = 128.0, unskipped = 0) and float(flat.loss_scaler.current[3]) == 0.0  # static: no growth window
	ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2')
	assert ca.train.amp_state_dict(opt) == dict(loss_scaler0 = dict(loss_scale = 65536.0, unskipped = 0))
	ca.train.amp_load_state_dict(opt, dict(loss_scaler0 = dict(loss_scale = 1024.0, unskipped = 17)))
	assert ca.train.amp_state_dict(opt) == dict(loss_scaler0 = dict(loss_scale = 1024.0, unskipped = 17))
	s = flat.loss_scaler
	a, b = s.pair()
	assert a.data_ptr() == s.current.data_ptr() and b.data_ptr() != a.data_ptr()
	s.advance()
	assert s.current.data_ptr() == b.data_ptr()


def test_gradient_buckets_are_graded_and_cover_the_arena():
	model = ca.models.Wav2Letter(64, [38])
	flat = ca.train.FlatParameters(model)
	engine = ca.parallel.DataParallelEngine(model, flat = flat)
	sizes = [(b['hi'] - b['lo']) * 4 / 2 ** 20 for b in engine.buckets]
	assert sizes[0] <= 4.5 and sizes[1] <= 8.5 and sizes[2] <= 16.5 and len(sizes) >= 6  # MiB: small where backward finishes last
	assert engine.buckets[0]['lo'] == 0 and all(a['hi'] <= b['lo'] for a, b in zip(engine.buckets, engine.buckets[1:]))
	assert sum(len(b['params']) for b in engine.buckets) == len(flat.params)
	assert engine.buckets[-1]['hi'] == flat.offsets[-1] + flat.params[-1].numel()


def test_fold2_pads_only_when_the_conv_output_is_unchanged():
	"""The zero frame in front of the stride-2 prologue (functional.Fold2.wants_even_input): only for an odd frame count, an odd kernel
	(the output length must not change) and channel counts inside the fold's envelope."""
	want = Fn.Fold2.wants_even_input
	assert want((256, 64, 11), ConvSpec(11, 2, 1, 5), 1501)
	assert not want((256, 64, 11), ConvSpec(11, 2, 1, 5), 1502)   # already even
	assert not want((256, 64, 12), ConvSpec(12, 2, 1, 6), 1501)   # even kernel: one more output frame with the padded input
	assert not want((256, 40, 11), ConvSpec(11, 2, 1, 5), 1501)   # 2 * Cin not a multiple of 128
	assert not want((200, 64, 11), ConvSpec(11, 2, 1, 5), 1501)   # Cout not a multiple of 128
	assert not want((256, 64, 11), ConvSpec(11, 1, 1, 5), 1501)   # not strided


def test_accepted_configuration_flags_build_the_reference_state_dict_layout():
	kw = dict(base_width = 32, kernel_sizes = [11, 13], out_width_factors = [2, 2], dropouts = [0.2, 0.2], out_width_factors_large = [2, 2], repeat = 2, num_subblocks = 1)
	plain = ca.models.JasperNet(64, [38], **kw)
	assert list(ca.models.JasperNet(64, [38], inplace = True, **kw).state_dict()) == list(plain.state_dict())
	sep = ca.models.JasperNet(64, [38], separable = True, groups = 16, **kw)
	keys = list(sep.state_dict())
	assert 'backbone.1.conv.0.0.bias' in keys and 'backbone.1.conv.0.2.weight' in keys and tuple(sep.state_dict()['backbone.1.conv.0.0.weight'].shape) == (64, 4, 11)
	assert 'backbone.0.conv.0.0.bias' not in keys  # the prologue is not separable (models.py:200-211)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window', stft_mode = 'conv')
	assert tuple(fe.state_dict()['stft.weight'].shape) == (514, 1, 512)
	with pytest.raises(ValueError):
		ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window', stft_mode = 'fft2')
	norm = ca.models.MaskedInstanceNorm1d(64, affine = False, track_running_stats = False, legacy = False)
	assert norm.legacy is False
	bpe = ca.models.JasperNet(64, [38, 48], decoder_type = 'bpe', **kw)
	assert any(k.startswith('decoder.1.1.bn.0') for k in bpe.state_dict())
	for name in ('JasperNetSeparable', 'JasperNetBigInplace', 'Wav2LetterDenseNoDilationInplace', 'JasperNetLarge', 'Wav2Letter'):
		assert hasattr(ca.models, name)


def test_ctc_workspace_query_follows_the_state_split():
	"""convasr_ctc_workspace_bytes is host code (no GPU call): lattice rows hold 128 (NPH + NPL) states, the smallest split that covers
	the 2 S + 1 states of the extended target -- (1,1) up to 127 labels, (2,1) to 191, ... (4,4) to 511, (5,4) ... (8,8) to 1,023; longer targets are refused (-1)."""
	from convasr_amd import _lib
	lib = _lib.load()
	B, T = 3, 753
	NB = T // 8 + 1
	for S, cap in [(0, 256), (1, 256), (127, 256), (128, 384), (150, 384), (191, 384), (192, 512), (255, 512), (256, 640), (319, 640), (320, 768), (383, 768), (384, 896), (447, 896), (448, 1024), (511, 1024), (512, 1152), (575, 1152), (576, 1280), (831, 1664), (832, 1792), (959, 1920), (960, 2048), (1023, 2048)]:
		want = (2 * B * T * cap + 4 * B * NB + 2 * B) * 4
		assert lib.convasr_ctc_workspace_bytes(B, T, S) == want, (S, cap)
	assert lib.convasr_ctc_workspace_bytes(B, T, 1024) == -1


def test_every_named_configuration_of_the_reference_builds_the_same_network():
	"""`getattr(models, args.model)` (train.py:428, transcribe.py:44): all 24 JasperNet subclasses of the reference's models.py (819-1442) exist
	here under the same names and build the same network -- per block the conv / batch-norm / residual-branch geometry, activation, dropout and
	mask flag, the residual policy, the feature normalisation's flags, every state-dict key and shape -- as the reference's own constructors did
	(tests/golden/model_zoo.json, written by make_golden_r5.py through tests/golden/describe_model.py)."""
	import json
	import sys
	import convasr_amd as ca
	sys.path.insert(0, GOLDEN)
	from describe_model import describe
	zoo = json.load(open(os.path.join(GOLDEN, 'model_zoo.json')))
	assert len(zoo) == 27
	for key, want in sorted(zoo.items()):
		name, _, variant = key.partition(':')
		cls = getattr(ca.models, name)
		kw = dict(base_width = 128 if 'Separable' in name else 16)
		args = (64, [38, 300]) if variant == 'bpe' else (64, [38])
		if variant == 'bpe':
			kw['decoder_type'] = 'bpe'
		got = json.loads(json.dumps(describe(cls(*args, **kw))))
		for field in want:
			assert got[field] == want[field], (key, field)


def test_split_network_wiring_decides_which_outputs_exist_as_planes_only():
	"""Round 6, host logic only: in a split-operand network (set_compute_dtype('bf16x3')) a layer output is handed on as its 16-bit planes ONLY when
	its one reader is a split conv -- the next repeat of the block, or the first conv of the block the network wired behind it (feeds_block);
	the last block (its reader is the decoder), tapped outputs of residual networks, evaluation mode and plain fp32 keep real fp32 outputs."""
	import torch
	import convasr_amd as ca
	flags = lambda m: [[blk._planes_out(r, r == len(blk.conv) - 1) for r in range(len(blk.conv))] for blk in m.backbone]
	m = ca.models.Wav2Letter(64, [38], compute_dtype = 'bf16x3').train()
	assert m.compute_dtype == torch.float32 and m.split_dtype == torch.bfloat16 and m.decoder.split_dtype == torch.bfloat16
	f = flags(m)
	assert all(all(row) for row in f[:-1]) and f[-1] == [False], f  # 17 of the 18 layer outputs exist as planes only; the last feeds the head
	assert [blk.feeds_block is not None for blk in m.backbone] == [True] * 7 + [False]
	m.eval()
	assert not any(any(row) for row in flags(m))
	m.train()
	m.set_compute_dtype(torch.float32)
	assert m.split_dtype is None and not any(any(row) for row in flags(m))
	m.set_compute_dtype('f16x3', inference = True)
	assert m.split_dtype == torch.float16 and all(blk.split_inference for blk in m.backbone) and all(all(row) for row in flags(m)[:-1])
	# a dense-residual network: block outputs that later blocks tap have several readers -> real tensors; repeats inside a block still hand planes on
	j = ca.models.JasperNet(64, [38], base_width = 64, kernel_sizes = [11, 13], out_width_factors = [2, 3], dropouts = [0.0, 0.0], out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, compute_dtype = 'bf16x3').train()
	fj = flags(j)
	tapped = [blk.tapped_output for blk in j.backbone]
	for blk, row, tap in zip(j.backbone, fj, tapped):
		assert row[:-1] == [True] * (len(row) - 1)  # inside a block: the next repeat is the one reader
		assert row[-1] == (not tap and blk.feeds_block is not None), (row, tap)
	assert any(tapped) and not fj[-1][-1]
	# state-dict keys are untouched by the wiring (feeds_block is a weak reference, not a registered submodule)
	assert not any('feeds_block' in k for k in j.state_dict())


def test_split_operand_arithmetic_restated_on_the_cpu():
	"""What the MI355X split-operand path claims, checked on the CPU against float64 (oracle.conv1d_split3, the restatement of its arithmetic): two
	bf16 planes hold 16 significant bits (fp16: 22, above its subnormal range), x - hi is exact, and a conv formed from the three kept
	products sits 2^-17 .. 2^-16 from the float64 conv -- three orders of magnitude inside plain bf16's 2^-9 -- with the dropped lo x lo
	term accounting for what is left."""
	import torch
	import torch.nn.functional as F
	from oracle import convasr_oracle as O
	torch.manual_seed(0)
	x = torch.randn(3, 64, 200) * torch.logspace(-2, 1, 64).view(1, 64, 1)
	w = torch.randn(96, 64, 11) / (64 * 11) ** 0.5
	for dt, bits in ((torch.bfloat16, 8), (torch.float16, 11)):
		hi, lo = O.split_planes(x, dt)
		assert torch.equal((x - hi).to(dt).float(), lo) and bool(((hi + lo - x).abs() <= x.abs() * 2.0 ** (-2 * bits) + (2.0 ** -25 if dt == torch.float16 else 0)).all())
	ref = F.conv1d(x.double(), w.double(), padding = 5)
	rel = lambda a: float((a.double() - ref).norm() / ref.norm())
	e3 = rel(O.conv1d_split3(x, w, padding = 5))
	e1 = rel(F.conv1d(x.bfloat16().double(), w.bfloat16().double(), padding = 5))
	assert 1e-6 < e3 < 8e-6 and e1 > 300 * e3, (e3, e1)  # measured on the MI355X kernels: 4.5e-6 and 2.3e-3
	xh, xl = O.split_planes(x); wh, wl = O.split_planes(w)
	full = O.conv1d_split3(x, w, padding = 5) + F.conv1d(xl.double(), wl.double(), padding = 5)  # + the dropped term: what is left is the planes' own residue
	assert rel(full) < 0.8 * e3
	assert rel(O.conv1d_split3(x, w, padding = 5, dtype = torch.float16)) < 5e-7
