"""torch.autograd.Function wrappers: each forward/backward is a fixed sequence of libconvasr_hip.so kernels.

Parameters keep the reference's fp32 (Cout, Cin, K) layout; the packed compute-dtype copies the MFMA kernels consume are
cached per parameter version.  When a parameter carries `_convasr_grad` (a pre-allocated fp32 gradient view installed by
`convasr_amd.train.FlatParameters`), backward writes / accumulates into it directly and hands autograd `None` -- no
extra gradient copies, and the data-parallel engine is told the moment each gradient is final.
"""
import os

import torch

from . import ops, _lib

_pack_cache = {}
_param_epoch = [0]  # bumped by optimizers that update parameters behind torch's back (convasr_amd.train.SGD)


def bump_param_epoch():
	_param_epoch[0] += 1


def param_version(w):
	"""What a packed copy of `w` is valid for: torch's in-place version counter, the storage address, and the epoch optimizers bump
	when they update parameters behind torch's back."""
	return (w._version, w.data_ptr(), _param_epoch[0])


def packed_weight(w, dtype, mode):
	"""Packed [K][rows_pad][cols] copy of a conv parameter, refreshed (in place) only when the parameter changed.  In training
	the forward and the dgrad layout are produced together the first time either is asked for after an update.
	A tap-major arena parameter (train.FlatParameters) whose Cout needs no row padding has no forward copy of its own: in fp32
	the master IS the packed operand, in bf16 / fp16 it is the parameter's segment of the arena's 16-bit mirror, which the fused optimizer
	kernels keep current -- only the transposed dgrad copy is still built by a packing launch."""
	ver = param_version(w)
	ent = _pack_cache.get((id(w), dtype))
	if ent is not None:
		# fast path (most calls of a step): the copy asked for is current.  For a weight served from the arena's 16-bit mirror that
		# also means: the mirror object is still the one the entry points into, and the arena vouches for this version of the segment
		if mode == _lib.PACK_DGRAD:
			if ent['dgr_ver'] == ver:
				return ent['dgr']
		elif ent['fwd_ver'] == ver:
			flat = ent.get('flat')
			if flat is None or (flat.data16 is ent.get('mirror') and flat._mirror_ver.get(id(w)) == ver):
				return ent['fwd']
	if ent is None:
		ent = _pack_cache[(id(w), dtype)] = dict(w = w, fwd = None, dgr = None, fwd_ver = None, dgr_ver = None)  # holds `w`: id() stays unique
	arena = getattr(w, '_convasr_arena', None)
	# (the address test: p.data may have been replaced since FlatParameters re-homed it -- model.to(), load_state_dict(assign = True) --
	# with the strides preserved; such a parameter no longer aliases the arena and is packed like any other)
	in_place = arena is not None and (ops.weight_layout(w) == _lib.W_KMAJOR or (w.shape[2] == 1 and w.is_contiguous())) and ops.cout_pad(w.shape[0]) == w.shape[0] and dtype in (torch.float32, ) + ops.HALF_DTYPES and w.device == arena[0].data.device and w.data_ptr() == arena[0].data.data_ptr() + 4 * arena[1]
	if in_place:
		flat, off = arena
		Cout, Cin, K = w.shape
		if dtype == torch.float32:
			ent['fwd'], ent['fwd_ver'] = flat.data[off:off + w.numel()].view(K, Cout, Cin), ver
		else:
			# the mirror segment is current only if the arena says so for THIS version: a mirror re-allocated in the other 16-bit type
			# (FlatParameters.mirror) is zero-filled, and a cache entry left from before the switch must not vouch for it
			ent['fwd'] = flat.mirror(dtype)[off:off + w.numel()].view(K, Cout, Cin)
			ent['fwd_ver'] = ver if flat._mirror_ver.get(id(w)) == ver else None
			ent['flat'], ent['mirror'] = flat, flat.data16  # (what the fast path above re-checks)
	if mode == _lib.PACK_FWD and ent['fwd_ver'] == ver:
		return ent['fwd']
	if mode == _lib.PACK_DGRAD and ent['dgr_ver'] == ver:
		return ent['dgr']
	if mode == _lib.PACK_DGRAD:
		ent['fwd'], ent['dgr'] = ops.pack_weight(w, dtype, None, out = (ent['fwd'], ent['dgr']), fwd_is_current = ent['fwd_ver'] == ver)
		ent['fwd_ver'] = ent['dgr_ver'] = ver
	else:  # (the forward copy alone, also in training: the dgrad copies of a step are made together, by prepack_dgrad_weights' one launch)
		ent['fwd'] = ops.pack_weight(w, dtype, _lib.PACK_FWD, out = (ent['fwd'], None))
		ent['fwd_ver'] = ver
	if in_place and dtype in ops.HALF_DTYPES:
		arena[0]._mirror_ver[id(w)] = ver
	return ent['fwd'] if mode == _lib.PACK_FWD else ent['dgr']


_structure_epoch = [0]


def invalidate_pack_cache():
	"""Called when conv modules were replaced (fuse_conv_bn_eval): packed copies are dropped and everything derived from the module tree
	(JasperNet's list of dgrad weights, captured step graphs) is rebuilt on next use."""
	_pack_cache.clear()
	_split_cache.clear()
	_structure_epoch[0] += 1


_split_cache = {}


def force_repack():
	"""Declare every packed copy that a launch of this module keeps current stale (the arena's 16-bit mirror segments, which the fused
	optimizer kernels write, stay vouched for): the next forward / backward re-packs them all.  train.GraphedTrainStep calls it right
	before a capture so that the graph records every per-step pack launch whatever ran since the last optimizer step."""
	for ent in _pack_cache.values():
		ent['dgr_ver'] = None
		if ent.get('flat') is None:
			ent['fwd_ver'] = None
	for cache in (Fold2._cache, _HeadPad._cache, _split_cache):
		for ent in cache.values():
			ent['ver'] = None


def split_weight(w, dtype, dgrad_planes = 3):
	"""(forward, dgrad) split operands of a conv parameter for the split-operand path (csrc/split3.hip): hi / lo planes of the fp32 master
	in the 16-bit type `dtype`, both refreshed by ONE launch when the parameter changed (in place: stable addresses).  dgrad_planes = 1: the
	dgrad operand is the ordinary 16-bit one (w_hi alone), for layers whose backward runs one product per gradient (cfg['split_hi_bwd'])."""
	ver = param_version(w)
	ent = _split_cache.get((id(w), dtype, dgrad_planes))
	if ent is None:
		ent = _split_cache[(id(w), dtype, dgrad_planes)] = dict(w = w, ver = None, fwd = None, dgr = None)  # holds `w`: id() stays unique
	if ent['ver'] != ver:
		ent['fwd'], ent['dgr'] = ops.pack_weight_split3(w, dtype, out = (ent['fwd'], ent['dgr']), dgrad_planes = dgrad_planes)
		ent['ver'] = ver
	return ent['fwd'], ent['dgr']


_PLANES_ATTR = '_convasr_planes_only'
_PLANES_CACHE_ATTR = '_convasr_planes_cache'  # on a REAL fp32 block output that several residual branches read: its planes, split once
_nan_cells = {}


def _planes_placeholder(z3, B, C, T):
	"""What a layer whose output exists as split-operand planes only (cfg['planes_out']) hands to autograd: a (B, C, T) fp32 tensor of the
	output's logical shape -- the consumer's dgrad returns a gradient of exactly that shape -- with NO memory behind it (a stride-0 view of one
	NaN) and the planes hanging on it.  The one reader the network wired behind the layer takes the planes off it (_take_planes); ops.as_cl
	refuses it, and any other arithmetic on it yields NaN rather than plausible numbers."""
	dev = z3.device
	cell = _nan_cells.get(dev)
	if cell is None:
		cell = _nan_cells[dev] = torch.full((1, ), float('nan'), dtype = torch.float32, device = dev)
	z = cell.as_strided((B, C, T), (0, 0, 0))
	setattr(z, _PLANES_ATTR, z3)
	return z


def _take_planes(x):
	"""Remove and return the split-operand planes the producer of `x` left on its placeholder (None for an ordinary tensor)."""
	d = getattr(x, '__dict__', None)
	return d.pop(_PLANES_ATTR, None) if d else None


def split_applies(split, dt, spec, Cin, Cout):
	"""Does a conv of this geometry run as a split-operand conv?  fp32 storage with a 16-bit plane type set, stride 1 and channel counts
	inside the LDS-DMA kernels' envelope (3 Cin % 64 == 0; the weight gradient's 128-channel tiles take Cin, Cout % 128 == 0 and fall back to the
	general 16-bit kernel otherwise).  The strided prologue joins through its stride-1 fold (Fold2.plan(split = ...)), a narrow one-tap head as a
	128-class problem (_HeadPad.split_weight), the one-tap residual branches of a dense block with the tapped output's planes shared by its readers; what
	fits none of these (channel counts outside the envelope) stays on the exact-fp32 kernels."""
	return split is not None and dt == torch.float32 and spec.stride == 1 and Cin % 64 == 0 and Cout % 8 == 0


def structure_epoch():
	return _structure_epoch[0]


# The transposed, tap-flipped dgrad copies of the weights (one ~8 us memory-bound launch per layer and step) depend on nothing but the
# parameters, and the step has one stretch where most of the chip idles: the CTC alpha / beta recursion, one workgroup per utterance
# (64 of 256 CUs, ~0.3 ms of dependent steps).  prepack_dgrad_weights() -- called by the network between the decoder and the loss --
# runs those launches on a side stream there; the first dgrad of the backward pass joins it (join_prepack).
PREPACK = os.environ.get('CONVASR_NO_PREPACK') != '1'  # A/B hook
GRAPHS_CAPTURED = [False]  # set by train.GraphedTrainStep at its first capture: from then on an eager step (the warm-up of a new batch shape, a shape beyond max_graphs) makes its dgrad copies on the MAIN stream -- see prepack_dgrad_weights
_prepack_streams = {}  # device -> [side stream, packs pending on it?]


GROUPED_PACK_MIN = int(os.environ.get('CONVASR_GROUPED_PACK_MIN', 32))
_pack_tables = {}  # (device, dtype, the (src, dst) addresses of the group) -> dict(items = device table, blocks, n); entries live as long as the process: graphs bake their addresses


def _pack_group(stale, dtype, refresh = True):
	"""Which of the `stale` weights' dgrad copies can share the one grouped launch: 16-bit, even channel counts, buffers allocated, packed
	forward copy current (the arena mirror's segment, or refreshed here when refresh is set)."""
	group = []
	# (a grouped launch only for MANY copies: the one launch floods every CU at once -- beside the CTC recursion, whose 64 polling workgroups
	# it overlaps on the prepack stream, that cost the Wav2Letter step +40 us of CTC time for 18 copies, where 18 small launches cost nothing;
	# JasperNetLarge's 108 copies drop from 0.83 to 0.24 ms: profiles/r05_ab_rounds_wav2letter.json, r05_config4_kernel_stats.csv)
	if dtype in ops.HALF_DTYPES and len(stale) >= GROUPED_PACK_MIN:
		for w in stale:
			ent = _pack_cache.get((id(w), dtype))
			if ent is None or ent['dgr'] is None:
				continue
			fwd = packed_weight(w, dtype, _lib.PACK_FWD) if refresh else ent['fwd']  # (no launch when the optimizer's 16-bit mirror serves it)
			Cout, Cin, K = w.shape
			if fwd is not None and Cout % 2 == 0 and Cin % 2 == 0 and fwd.dtype == dtype:
				group.append((w, ent, fwd))
	return group


def _pack_table(group, dtype):
	"""The device-resident item table of a group (rebuilt -- one small host-to-device copy -- only when the set of buffers changed)."""
	import struct
	dev = group[0][0].device
	key = tuple((fwd.data_ptr(), ent['dgr'].data_ptr()) for _, ent, fwd in group)
	tab = _pack_tables.get((dev, dtype, key))  # one table per set of buffers, never dropped: a captured step graph holds its address (a few KB each)
	if tab is None:
		if capturing():
			raise _lib.ConvasrHipError('the dgrad pack table changed while a step graph is being captured (GraphedTrainStep prewarms it: functional.prewarm_dgrad_pack)')
		assert _lib.load().convasr_pack_dgrad_item_bytes() == 40
		blob, first = b'', 0
		for w, ent, fwd in group:
			Cout, Cin, K = w.shape
			blob += struct.pack('<QQiiiiii', fwd.data_ptr(), ent['dgr'].data_ptr(), Cout, Cin, K, ops.cout_pad(Cout), ops.cout_pad(Cin), first)
			first += K * ((Cout + 63) // 64) * ((Cin + 63) // 64)
		tab = _pack_tables[(dev, dtype, key)] = dict(key = key, items = torch.frombuffer(bytearray(blob), dtype = torch.uint8).to(dev), blocks = first, n = len(group))
	return tab


def prewarm_dgrad_pack(weights, dtype):
	"""Build the grouped pack's device table for the weights a training step will re-pack (all of them: every optimizer step makes every
	copy stale), outside a graph capture -- the capture itself cannot copy a table to the device."""
	group = _pack_group([w for w in weights], dtype, refresh = False)
	if group:
		_pack_table(group, dtype)


def _pack_dgrad_many(stale, dtype):
	"""The transposed dgrad copies of the `stale` weights: ONE launch for all those whose packed forward copy is current and 16-bit (the
	arena mirror's segments, or a forward pack refreshed here), a launch each for the rest."""
	group = _pack_group(stale, dtype)
	grouped = {id(w) for w, _, _ in group}
	for w in stale:
		if id(w) not in grouped:
			packed_weight(w, dtype, _lib.PACK_DGRAD)
	if not group:
		return
	tab = _pack_table(group, dtype)
	_lib.call('convasr_pack_dgrad_grouped', _lib.ptr(tab['items']), tab['n'], tab['blocks'], _lib.stream_ptr())
	for w, ent, _ in group:
		ent['dgr_ver'] = param_version(w)


def prepack_dgrad_weights(weights, dtype):
	if not weights:
		return
	dev = weights[0].device
	join_prepack(dev)  # (a previous step that never reached its backward pass -- a loss skipped on the host -- left its packs unjoined)
	stale = []
	for w in weights:
		ent = _pack_cache.get((id(w), dtype))
		if ent is None or ent['dgr'] is None:
			packed_weight(w, dtype, _lib.PACK_DGRAD)  # first use: buffers are allocated (and stay owned) by the main stream
		elif ent['dgr_ver'] != param_version(w):
			stale.append(w)
	if not stale:
		return
	# Once step graphs exist in this process, eager steps stay off the prepack stream too: one stream less at the boundary between an eagerly
	# launched step and a replayed one, where round 6 found a race (profiles/r06_interleave_race.txt).  This alone did not close it -- what does
	# is the host wait per transition in train.GraphedTrainStep._fence_transition -- but it removed the variant in which the replayed step's
	# pack nodes rewrote buffers an eager step had just packed on the side stream (scratch/r6_interleave_debug2.py: differing losses in every other
	# run before, none in 8 after).  Eager steps are the exception once graphs exist, so they simply do not fork.
	if not PREPACK or GRAPHS_CAPTURED[0]:  # (A/B hook, and what a linear step-graph capture sets: the copies are made on the main stream, still in one launch)
		_pack_dgrad_many(stale, dtype)
		return
	st = _prepack_streams.get(dev)
	if st is None:
		st = _prepack_streams[dev] = [torch.cuda.Stream(device = dev), False]
	side, main = st[0], torch.cuda.current_stream(dev)
	_lib.stream_wait(side, main)  # the parameters, their 16-bit mirror and every earlier reader of the buffers are ordered before the packs
	with torch.cuda.stream(side):
		_pack_dgrad_many(stale, dtype)
	st[1] = True


def join_prepack(device):
	st = _prepack_streams.get(device)
	if st is not None and st[1]:
		_lib.stream_wait(torch.cuda.current_stream(device), st[0])
		st[1] = False


CAPTURING = ops._capturing  # [False]; set by train.GraphedTrainStep while a training step is being captured into a HIP graph: every buffer a captured kernel touches must then be persistent or allocated inside the capture (no module-level / per-stream caches), and double-buffered device state is handed back instead of swapped


def capturing():
	return CAPTURING[0]


class _DropoutState:
	"""The dropout generator's host side.  A mask is a function of (seed, offset, element index) XOR a per-step key word that lives on
	the device (include/convasr_hip.h, convasr_step_begin): `offset` numbers the layers of ONE step (begin_step resets it), the key
	numbers the steps -- so a step replayed from a captured graph, whose seed / offset arguments are frozen, still draws fresh masks,
	and draws the same ones as the eager step it was captured from."""
	seed = 0x5EEDC0DE
	offset = 0
	dev = {}  # device -> dict(state = int64 (4,) device tensor {seed, steps begun, key of the current step, reserved}, seed = the seed it was initialised from)

	@classmethod
	def next(cls, numel):
		off = cls.offset
		cls.offset += (numel + 3) // 4 + 1
		return cls.seed, off

	@classmethod
	def key(cls, device):
		"""Device address of the current step's key word (what the kernels take as `step_key`), or None before the first begin_step()."""
		st = cls.dev.get(device)
		return None if st is None else st['state'].data_ptr() + 16

	@classmethod
	def upload(cls, device):
		signed = cls.seed - (1 << 64) if cls.seed >= (1 << 63) else cls.seed
		st = cls.dev.get(device)
		if st is None:
			st = cls.dev[device] = dict(state = torch.zeros(4, dtype = torch.int64, device = device), seed = None)
		st['state'].copy_(torch.tensor([signed, 0, 0, 0], dtype = torch.int64))
		st['seed'] = cls.seed
		return st


def manual_seed(seed):
	_DropoutState.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
	_DropoutState.offset = 0
	for device in list(_DropoutState.dev):
		_DropoutState.upload(device)  # (eagerly: a captured step graph reads the words, the host cannot patch them at the next begin_step)


def begin_step(device):
	"""First call of a training step (train.train_step): advances the device-resident step key (one single-thread launch) and restarts
	the per-layer offsets, so that step s, layer l draws the same mask whether the step is launched kernel by kernel or replayed."""
	device = torch.device(device)
	st = _DropoutState.dev.get(device)
	if st is None or st['seed'] != _DropoutState.seed:
		if CAPTURING[0]:
			raise _lib.ConvasrHipError('begin_step: the dropout state of this device must exist before a step is captured (run one eager step first)')
		st = _DropoutState.upload(device)
	_lib.call('convasr_step_begin', st['state'].data_ptr(), _lib.stream_ptr())
	_DropoutState.offset = 0


def _deliver(params, compute):
	"""Gradient hand-off for a group of parameters produced by one kernel sequence.

	compute(outs, accumulate) must fill outs[i] (fp32, parameter-shaped; None for parameters that need no gradient).
	If every live parameter has a gradient arena the kernels write / accumulate there and autograd gets None;
	otherwise fresh tensors are returned for autograd to accumulate."""
	live = [p is not None and p.requires_grad for p in params]
	if not any(live):
		return [None] * len(params)
	arenas = [getattr(p, '_convasr_grad', None) if ok else None for p, ok in zip(params, live)]
	if all(a is not None for a, ok in zip(arenas, live) if ok):
		fresh = {bool(getattr(p, '_convasr_fresh', True)) for p, ok in zip(params, live) if ok}
		if len(fresh) != 1:
			raise _lib.ConvasrHipError('gradient arenas of one layer are out of step (mixed fresh / accumulated state)')
		compute(arenas, not fresh.pop())
		for p, ok in zip(params, live):
			if ok:
				p._convasr_fresh = False
				hook = getattr(p, '_convasr_ready', None)
				if hook is not None:
					hook(p)
		return [None] * len(params)
	outs = [torch.empty_like(p, dtype = torch.float32, memory_format = torch.contiguous_format) if ok else None for p, ok in zip(params, live)]
	compute(outs, False)
	return outs


def _deliver_many(groups, compute):
	"""_deliver for several parameter groups (e.g. the (gamma, beta) pairs of a dense block's batch norms) filled by ONE launch:
	compute(outs, accs) with outs[i] the list of fp32 outputs of group i (None where no gradient is wanted) and accs[i] its accumulate
	flag.  Returns per group what autograd should get (None for arena parameters)."""
	outs, accs, ret, arena_groups = [], [], [], []
	for params in groups:
		live = [p is not None and p.requires_grad for p in params]
		arenas = [getattr(p, '_convasr_grad', None) if ok else None for p, ok in zip(params, live)]
		if any(live) and all(a is not None for a, ok in zip(arenas, live) if ok):
			fresh = {bool(getattr(p, '_convasr_fresh', True)) for p, ok in zip(params, live) if ok}
			if len(fresh) != 1:
				raise _lib.ConvasrHipError('gradient arenas of one layer are out of step (mixed fresh / accumulated state)')
			outs.append(arenas); accs.append(not fresh.pop()); ret.append([None] * len(params)); arena_groups.append((params, live))
		else:
			o = [torch.empty_like(p, dtype = torch.float32, memory_format = torch.contiguous_format) if ok else None for p, ok in zip(params, live)]
			outs.append(o); accs.append(False); ret.append(o)
	compute(outs, accs)
	for params, live in arena_groups:
		for p, ok in zip(params, live):
			if ok:
				p._convasr_fresh = False
				hook = getattr(p, '_convasr_ready', None)
				if hook is not None:
					hook(p)
	return ret


GROUP_RES = os.environ.get('CONVASR_NO_GROUPED_RES') != '1'  # A/B hook: a dense block's residual branches in grouped launches + gradient accumulators on the tapped outputs
_GACC_ATTR = '_convasr_gacc'
RES_WGRAD_SIDE = os.environ.get('CONVASR_NO_RES_WGRAD_SIDE') != '1'  # A/B hook: the residual branches' weight gradients on the wgrad side stream too
after_long_launch_hooks = {}  # id -> callable, run right after the backward pass has enqueued a long kernel (a dgrad): the data-parallel engine enqueues its ready collectives there, where their host cost hides behind queued GPU work
_side_streams = {}  # device -> torch.cuda.Stream running the weight-gradient kernels (None entry = disabled)


def enable_side_stream_wgrad(device, enabled = True):
	"""Run every wgrad (+ its split-K reduce) on a second HIP stream so its workgroups fill the CUs the concurrent dgrad /
	BN-backward kernels of the main stream leave idle (partial last rounds, waits).  Only meaningful with gradient arenas
	(FlatParameters): the consumer of the gradients must call join_side_streams() first (train_step does)."""
	device = torch.device(device)
	_side_streams[device] = torch.cuda.Stream(device = device) if enabled else None


def side_stream(device):
	"""The wgrad side stream of `device`, or None when it is disabled."""
	return _side_streams.get(torch.device(device))


SIDE_KEEPALIVE = os.environ.get('CONVASR_NO_SIDE_KEEPALIVE') != '1'  # the side stream's operands are kept alive until the join instead of record_stream() (round 6: -0.6 .. -0.8 % on the JasperNetLarge step, +0.4 GiB; profiles/r06_ab_side_keepalive.txt)
_side_keepalive = {}


def join_side_streams():
	for dev, side in _side_streams.items():
		if side is not None:
			_lib.stream_wait(torch.cuda.current_stream(dev), side)
	_wgrad_pending.clear()
	for held in _side_keepalive.values():
		del held[:]  # (released on the main stream, which has just been ordered behind the side stream's last reader)


# WGRAD_AFTER_DGRAD (round 6, with the side stream on): a layer's weight gradient is enqueued on the side stream AFTER the same layer's dgrad and
# ordered behind it, and the NEXT dgrad of the backward pass waits for it: the two MFMA-bound kernels of a layer never run side by side (that cost
# the Wav2Letter step 1.5 % in round 3), but the memory-bound BN passes of the next layer (reduce / finalize / apply) run UNDER the weight gradient
# instead of after it.  Off: the round-3 form -- wgrad enqueued before dgrad, both concurrently.
WGRAD_AFTER_DGRAD = os.environ.get('CONVASR_WGRAD_AFTER_DGRAD', '0') == '1'
_wgrad_pending = {}  # device -> a side-stream weight gradient the next dgrad must wait for


def _join_pending_wgrad(dev):
	if _wgrad_pending.pop(dev, False):
		side = _side_streams.get(dev)
		if side is not None:
			torch.cuda.current_stream(dev).wait_stream(side)


def _run_wgrad(dev, tensors, fn):
	side = _side_streams.get(dev)
	if side is None:
		return fn()
	if WGRAD_AFTER_DGRAD:
		_wgrad_pending[dev] = True
	_lib.stream_wait(side, torch.cuda.current_stream(dev))
	with torch.cuda.stream(side):
		out = fn()
	held = _side_keepalive.setdefault(dev, [])
	if SIDE_KEEPALIVE and len(held) < 16384:  # (a caller that never joins must not pile tensors up: past this many the allocator's own bookkeeping takes over)
		held.extend(tensors)  # held until join_side_streams(): no record_stream bookkeeping (an allocator event per tensor at its release)
	else:
		for t in tensors:
			t.record_stream(side)
	return out


SPLIT_FAMILY, SPLIT_WGRAD_FAMILY = 'conv1d_igemm_v2s_kernel<x3>', 'conv1d_wgrad<x3>'  # the bench's per-kernel timer books the split-operand launches apart (algorithmic FLOPs, three MFMAs each)


class ConvSpec:
	"""Static description of one Conv1d (+ optional BatchNorm) as the kernels need it."""

	def __init__(self, K, stride = 1, dilation = 1, padding = 0):
		self.K, self.stride, self.dilation, self.padding = K, stride, dilation, padding


class Fold2:
	"""The stride-2 prologue conv (models.py:312) as a stride-1 conv over the (T / 2, 2 Cin) view of its input (include/convasr_hip.h,
	"Stride-2 fold"): the forward and the weight gradient then run in the LDS-DMA kernels like every other layer, where the general
	register-staged kernels took 73 + 161 us per step for 7 % of one big layer's FLOPs.  Needs an even number of input frames (the
	instance norm in front of the prologue pads its output by one zero frame for this, ops.instnorm(pad_time_to = 2)) and no input
	gradient (a strided dgrad does not exist here)."""
	_cache = {}
	enabled = os.environ.get('CONVASR_NO_FOLD2') != '1'  # tests / A-B runs flip this

	@classmethod
	def plan(cls, x, weight, spec, dt, x_needs_grad, split = None):
		"""None, or (view of x, K', P', Tout) when the fold applies.  split (a 16-bit plane type) with dt = fp32: the folded conv of a
		split-operand network (its view is split into planes by the caller)."""
		B, Cin, Tin = x.shape
		if not (cls.enabled and (dt in ops.HALF_DTYPES or (dt == torch.float32 and split is not None)) and spec.stride == 2 and spec.dilation == 1 and Tin % 2 == 0 and (2 * Cin) % 128 == 0 and weight.shape[0] % 128 == 0 and not x_needs_grad and ops.is_cl(x) and x.stride(0) == Tin * Cin):
			return None
		Kf, Pf = ops.fold2_geometry(spec.K, spec.padding)
		Tout = ops.conv_out_len(Tin, spec.K, 2, 1, spec.padding)
		if Tout > ops.conv_out_len(Tin // 2, Kf, 1, 1, Pf):
			return None
		return x.as_strided((B, 2 * Cin, Tin // 2), (Tin * Cin, 1, 2 * Cin)), Kf, Pf, Tout

	@classmethod
	def wants_even_input(cls, weight_shape, spec, Tin):
		"""Should an odd-length input (Tin frames) of this conv get one zero frame appended so that plan() applies?  Only when the conv's
		own output is unchanged by it (the extra frame stays inside the zero padding: the output length is the same, true for odd K
		with padding K // 2) and the fold's channel / geometry envelope holds."""
		Cout, Cin, K = weight_shape
		if not (cls.enabled and spec.stride == 2 and spec.dilation == 1 and Tin % 2 == 1 and (2 * Cin) % 128 == 0 and Cout % 128 == 0):
			return False
		Tout = ops.conv_out_len(Tin, spec.K, 2, 1, spec.padding)
		if ops.conv_out_len(Tin + 1, spec.K, 2, 1, spec.padding) != Tout:
			return False
		Kf, Pf = ops.fold2_geometry(spec.K, spec.padding)
		return Tout <= ops.conv_out_len((Tin + 1) // 2, Kf, 1, 1, Pf)

	@classmethod
	def packed_weight(cls, weight, dt, pad):
		ver = param_version(weight)
		ent = cls._cache.get((id(weight), dt))
		if ent is None:
			ent = cls._cache[(id(weight), dt)] = dict(w = weight, ver = None, wp = None)
		if ent['ver'] != ver:
			ent['wp'] = ops.fold2_pack_weight(weight, dt, pad, out = ent['wp'])
			ent['ver'] = ver
		return ent['wp']

	@classmethod
	def split_weight(cls, weight, split, pad):
		"""Forward planes [K'][Cout][3 x 2 Cin] of the folded conv of a split-operand network: the fp32 folded weight (convasr_fold2_pack_weight
		into a tap-major fp32 buffer) split like any other (functional.split_weight's kernel); refreshed per parameter version, in place."""
		ver = param_version(weight)
		ent = cls._cache.get((id(weight), 'split', split))
		if ent is None:
			ent = cls._cache[(id(weight), 'split', split)] = dict(w = weight, ver = None, wf = None, fwd = None)
		if ent['ver'] != ver:
			ent['wf'] = ops.fold2_pack_weight(weight, torch.float32, pad, out = ent['wf'])  # [K'][Cout][2 Cin] fp32 (Cout % 128 == 0: no padded rows)
			ent['fwd'] = ops.pack_weight_split3(ent['wf'].permute(1, 2, 0), split, out = (ent['fwd'], None), want_dgrad = False)[0]
			ent['ver'] = ver
		return ent['fwd']

	@staticmethod
	def wgrad_split(x3, dy3, weight, spec, Kf, Pf, out, accumulate):
		"""Weight gradient of the folded split conv: the plane tensors read as 3 T frames, dilation and padding tripled, then unfolded."""
		Cout, Cin, K = weight.shape
		xf, dyf = ops.split3_frames(x3), ops.split3_frames(dy3)
		B, Tout = dy3.shape[0], dy3.shape[2]
		dwf = torch.empty(Kf, Cout, 2 * Cin, dtype = torch.float32, device = dy3.device)
		ops.conv1d_wgrad(xf, dyf, Cout, Kf, 1, 3, 3 * Pf, dwf.permute(1, 2, 0), work = 2.0 * B * Tout * Cout * Cin * K, family = SPLIT_WGRAD_FAMILY)
		ops.fold2_unfold_wgrad(dwf, out, spec.padding, accumulate = accumulate)

	@staticmethod
	def wgrad_hi(x3, dy, weight, spec, Kf, Pf, out, accumulate):
		"""... with one product: plane 0 of the folded view's planes (read in place) against the dense 16-bit dy."""
		Cout, Cin, K = weight.shape
		B, _, Tout = dy.shape
		dwf = torch.empty(Kf, Cout, 2 * Cin, dtype = torch.float32, device = dy.device)
		ops.conv1d_wgrad_hi(x3, dy, Cout, Kf, 1, Pf, dwf.permute(1, 2, 0), work = 2.0 * B * Tout * Cout * Cin * K)
		ops.fold2_unfold_wgrad(dwf, out, spec.padding, accumulate = accumulate)

	@staticmethod
	def wgrad(xv, dy, weight, spec, Kf, Pf, out, accumulate):
		Cout, Cin, K = weight.shape
		B, _, Tout = dy.shape
		dwf = torch.empty(Kf, Cout, 2 * Cin, dtype = torch.float32, device = dy.device)
		ops.conv1d_wgrad(xv, dy, Cout, Kf, 1, 1, Pf, dwf.permute(1, 2, 0), work = 2.0 * B * Tout * Cout * Cin * K)
		ops.fold2_unfold_wgrad(dwf, out, spec.padding, accumulate = accumulate)


# Cross-layer backward fusion (bf16 / fp16 training): pass 1 of a layer's batch-norm backward (per-channel sums of g and g * xhat) runs
# in the epilogue of the dgrad launch that PRODUCES that layer's dz, i.e. in the backward of the layer's consumer.  No global
# state: the producer's forward hangs a `link` (what the epilogue needs) on its output tensor and keeps it on its autograd ctx;
# the consumer's forward takes the link off its input (so exactly one consumer can hold it) and keeps it on ITS ctx; the
# consumer's backward runs the fused dgrad and leaves the dz it returns in the link; the producer's backward skips its own
# reduce pass only if the gradient autograd hands it is that very tensor (same storage: nothing was accumulated or copied in
# between).  Only outputs with exactly one consumer get a link (ConvBn1d.single_consumer_output).
FUSE_BWD = os.environ.get('CONVASR_NO_BWD_FUSION') != '1'  # tests / A-B runs flip this to compare against the separate reduce pass
GATE_BITS = os.environ.get('CONVASR_NO_GATE_BITS') != '1'  # likewise: the stored one-bit gradient gates vs re-deriving them in backward
_LINK_ATTR = '_convasr_bwd_link'


def _take_link(x):
	"""Remove and return the fusion link the producer of `x` left on it (None if there is none)."""
	d = getattr(x, '__dict__', None)
	return d.pop(_LINK_ATTR, None) if d else None


def _bwd_sums_buffer(bn, C, dev, B, T):
	return _stats_buffer(bn, C, dev, B, T, slot = '_convasr_bwd_stats')


def _bn_backward_from_g(g, y, gamma, beta, bnp, sums, n):
	"""Pass 2 of a batch-norm backward whose g (gradient at the BN output) is materialised: a per-channel finalize turns the two
	sums into (dgamma, dbeta) and the coefficients of dy = A*g + Bc*y + D, then one streaming kernel applies them."""
	coef = torch.empty(3 * y.shape[1], dtype = torch.float32, device = y.device)
	finalize = lambda outs, acc: ops.bn_bwd_finalize(sums, gamma, bnp[0], bnp[1], n, coef = coef, dgamma = outs[0], dbeta = outs[1], accumulate = acc)
	if (gamma is not None and gamma.requires_grad) or (beta is not None and beta.requires_grad):
		dgamma, dbeta = _deliver([gamma, beta], finalize)
	else:
		finalize([None, None], False)
		dgamma = dbeta = None
	return dgamma, dbeta, ops.bn_act_bwd_apply(g, y, coef, False)


def _after_long_launch():
	for hook in after_long_launch_hooks.values():
		hook()


def _dgrad(x, dy, weight, spec, dt, link = None, wd = None, split = None, hi = False):
	"""dx of one conv; fused with the BN backward reduce of the layer that produced x when that layer left a link for it.
	wd: packed dgrad weights to use instead of the cached copy of `weight` (the channel-padded head, see _HeadPad).
	split: the 16-bit plane type of a split-operand conv -- dy is then its (B, 3 Cout, T) plane tensor (ops.split3, SPLIT_GRAD) and dx is fp32;
	hi: dy is the dense 16-bit (B, Cout, T) tensor instead and the product is dy x w_hi alone (cfg['split_hi_bwd']), dx still fp32."""
	pad = spec.dilation * (spec.K - 1) - spec.padding
	_join_pending_wgrad(dy.device)
	if split is not None:
		Cout, Cin, K = weight.shape
		B, _, Tdy = dy.shape
		if hi:
			dx = ops.conv1d(dy, split_weight(weight, split, 1)[1], Cin, spec.K, 1, spec.dilation, pad, out_dtype = torch.float32)
		else:
			dx = ops.conv1d(dy, split_weight(weight, split)[1], Cin, spec.K, 1, spec.dilation, pad, out_dtype = torch.float32, work = 2.0 * B * ops.conv_out_len(Tdy, K, 1, spec.dilation, pad) * Cout * Cin * K, family = SPLIT_FAMILY)
		_after_long_launch()
		return dx
	Cin = x.shape[1]
	join_prepack(dy.device)
	wd = packed_weight(weight, dt, _lib.PACK_DGRAD) if wd is None else wd
	if link is not None and dt in ops.HALF_DTYPES and spec.stride == 1:
		dx = ops.conv1d_dgrad_bn_reduce(dy, wd, Cin, spec.K, spec.dilation, pad, link['y'], link['bnp'][2], link['bnp'][3], link['bnp'][0], link['bnp'][1], link['act'], link['drop'][0], link['drop'][1], link['drop'][2], link['xl'], link['sums'], gate = link.get('gate'), step_key = link['drop'][3])
		if dx is not None:
			link['dz'] = dx  # held until the producer's backward has looked at it: the address cannot be recycled meanwhile
			_after_long_launch()
			return dx
	dx = ops.conv1d(dy, wd, Cin, spec.K, 1, spec.dilation, pad)
	_after_long_launch()
	return dx


class ConvBnActFunction(torch.autograd.Function):
	"""One repeat of ConvBn1d (models.py:128-138): conv -> BN(train) -> + sum BN(conv1x1(residual)) -> act -> dropout -> mask.

	apply(cfg, x, weight, gamma, beta, xlen, *flat_residuals) where flat_residuals is, per residual,
	(res_x, res_weight, res_bias, res_gamma, res_beta) -- the last four None for an identity ('flat') residual.
	cfg: dict(spec, bn (module holding running stats / momentum / eps), res_bn (list of modules or None), act, dropout_p,
	temporal_mask, compute_dtype)."""

	@staticmethod
	def forward(ctx, cfg, x, weight, gamma, beta, xlen, *flat_res):
		spec, dt = cfg['spec'], cfg['compute_dtype']
		ctx.producer_link = _take_link(x)
		planes = _take_planes(x)  # the producer wrote its output as split-operand planes only: x itself is a placeholder of the logical shape
		if planes is None:
			x = ops.as_cl(x, dt)
		B, Cin, Tin = x.shape
		Cout = weight.shape[0]
		dev = x.device
		act = cfg['act']
		xl = ops.xlen_f32(xlen, dev) if (cfg['temporal_mask'] and xlen is not None) else None
		n_res = len(flat_res) // 5
		if planes is not None and not (split_applies(cfg.get('split'), dt, spec, Cin, Cout) and planes.dtype == cfg['split'] and tuple(planes.shape) == (B, 3 * Cin, Tin)):
			raise _lib.ConvasrHipError('a layer output that exists as split-operand planes only reached a conv that does not run as a split conv (models.ConvBn1d._planes_out and this layer disagree)')

		bn = cfg['bn']
		stats = _stats_buffer(bn, Cout, dev, B, ops.conv_out_len(Tin, spec.K, spec.stride, spec.dilation, spec.padding))
		x_needs_grad = x.requires_grad or ctx.needs_input_grad[1]
		ctx.fold = Fold2.plan(x, weight, spec, dt, x_needs_grad, split = cfg.get('split')) if planes is None else None
		ctx.split = None
		if ctx.fold is not None and dt == torch.float32:
			# the strided prologue of a split-operand network: its stride-1 fold, with the folded view split into planes
			xv, Kf, Pf, Tout = ctx.fold
			ctx.split = cfg['split']
			x = ops.split3(xv, ctx.split, ops.SPLIT_INPUT)
			y = ops.conv1d(x, Fold2.split_weight(weight, ctx.split, spec.padding), Cout, Kf, 1, 1, Pf, out_dtype = torch.float32, stats = stats, Tout = Tout, work = 2.0 * B * Tout * Cout * Cin * spec.K, family = SPLIT_FAMILY)
			ctx.fold = (Kf, Pf)
		elif ctx.fold is not None:
			xv, Kf, Pf, Tout = ctx.fold
			y = ops.conv1d(xv, Fold2.packed_weight(weight, dt, spec.padding), Cout, Kf, 1, 1, Pf, stats = stats, Tout = Tout, work = 2.0 * B * Tout * Cout * Cin * spec.K)
			ctx.fold = (Kf, Pf)
		elif split_applies(cfg.get('split'), dt, spec, Cin, Cout):
			# split-operand conv (csrc/split3.hip): the fp32 input as three 16-bit planes per frame, read by the LDS-DMA kernel as 3 Cin channels;
			# the planes, not x, are what backward keeps (the weight gradient reads the same memory as 3 Tin frames of Cin channels)
			ctx.split = cfg['split']
			x = planes if planes is not None else ops.split3(x, ctx.split, ops.SPLIT_INPUT)
			y = ops.conv1d(x, split_weight(weight, ctx.split, 1 if cfg.get('split_hi_bwd') else 3)[0], Cout, spec.K, 1, spec.dilation, spec.padding, out_dtype = torch.float32, stats = stats, work = 2.0 * B * ops.conv_out_len(Tin, spec.K, 1, spec.dilation, spec.padding) * Cout * Cin * spec.K, family = SPLIT_FAMILY)
		else:
			y = ops.conv1d(x, packed_weight(weight, dt, _lib.PACK_FWD), Cout, spec.K, spec.stride, spec.dilation, spec.padding, stats = stats)
		Tout = y.shape[2]
		bnp = ops.bn_finalize(stats, B * Tout, gamma, beta, bn.running_mean, bn.running_var, _momentum(bn), bn.eps, num_batches_tracked = bn.num_batches_tracked)

		res_y, res_bnp, res_split = [], [], [False] * n_res
		# where each batch-normed branch's input gradient goes in backward: the gradient accumulator its producer left on the tapped tensor
		# (GRAD_ACC below), or None = hand it to autograd
		ctx.res_gacc = [getattr(flat_res[5 * r], _GACC_ATTR, None) if (GROUP_RES and flat_res[5 * r + 1] is not None) else None for r in range(n_res)]
		res_x = [ops.as_cl(flat_res[5 * r], dt) for r in range(n_res)]
		branches = [r for r in range(n_res) if flat_res[5 * r + 1] is not None]
		ys = None
		if GROUP_RES and len(branches) >= 2 and dt in ops.HALF_DTYPES:
			# all of the block's 1x1 residual convs (+ bias, + BN statistics) in ONE dispatch: same values as a launch each
			sts = [_stats_buffer(cfg['res_bn'][r], Cout, dev, B, Tout) for r in branches]
			ys = ops.conv1x1_grouped([res_x[r] for r in branches], [packed_weight(flat_res[5 * r + 1], dt, _lib.PACK_FWD) for r in branches], [Cout] * len(branches), biases = [flat_res[5 * r + 2] for r in branches], stats = sts)
			if ys is not None:
				ys, sts = dict(zip(branches, ys)), dict(zip(branches, sts))
		fin = None
		if ys is not None:  # ... and their batch norms' finalize (statistics -> mean / invstd / scale / shift, running statistics) in one launch
			rbns = [cfg['res_bn'][r] for r in branches]
			fin = dict(zip(branches, ops.bn_finalize_grouped([sts[r] for r in branches], B * Tout, [flat_res[5 * r + 3] for r in branches], [flat_res[5 * r + 4] for r in branches], [m.running_mean for m in rbns], [m.running_var for m in rbns],
				[_momentum(m) for m in rbns], [m.eps for m in rbns], [m.num_batches_tracked for m in rbns])))
		for r in range(n_res):
			rx, rw, rb, rg, rbeta = flat_res[5 * r:5 * r + 5]
			rx = res_x[r]
			if rw is None:
				res_y.append(rx)
				res_bnp.append(None)
			elif fin is not None:
				res_y.append(ys[r])
				res_bnp.append(fin[r])
			else:
				rbn = cfg['res_bn'][r]
				if ys is not None:
					ry, st = ys[r], sts[r]
				elif split_applies(cfg.get('split'), dt, ConvSpec(1), rx.shape[1], Cout):
					# a one-tap residual branch of a split-operand network: the tapped block output's planes (made once per step and kept on the
					# tensor: up to ten later blocks read the same output) against the branch weight's planes
					st = _stats_buffer(rbn, Cout, dev, B, Tout)
					cache = flat_res[5 * r].__dict__
					rx3 = cache.get(_PLANES_CACHE_ATTR)
					if rx3 is None or rx3.dtype != cfg['split']:
						rx3 = cache[_PLANES_CACHE_ATTR] = ops.split3(rx, cfg['split'], ops.SPLIT_INPUT)
					ry = ops.conv1d(rx3, split_weight(rw, cfg['split'], 1 if cfg.get('split_hi_bwd') else 3)[0], Cout, 1, 1, 1, 0, out_dtype = torch.float32, bias = rb, stats = st, work = 2.0 * B * Tout * Cout * rx.shape[1], family = SPLIT_FAMILY)
					res_x[r] = rx3  # (what backward keeps of this branch's input: the weight gradient reads the planes as 3 T frames)
					res_split[r] = True
				else:
					st = _stats_buffer(rbn, Cout, dev, B, Tout)
					ry = ops.conv1d(rx, packed_weight(rw, dt, _lib.PACK_FWD), Cout, 1, 1, 1, 0, bias = rb, stats = st)
				res_y.append(ry)
				res_bnp.append(ops.bn_finalize(st, B * Tout, rg, rbeta, rbn.running_mean, rbn.running_var, _momentum(rbn), rbn.eps, num_batches_tracked = rbn.num_batches_tracked))

		p_drop = cfg['dropout_p']
		seed, offset = _DropoutState.next(B * Cout * Tout) if p_drop > 0 else (0, 0)
		skey = _DropoutState.key(dev) if p_drop > 0 else None
		# one bit per element: does the gradient pass it (activation range, dropout, frame mask)?  The backward kernels of a residual-free
		# layer take the bits back in instead of re-deriving the pre-activation, re-hashing the dropout mask and redoing the frame arithmetic
		gate = None
		if GATE_BITS and (n_res == 0 or (GROUP_RES and dt in ops.HALF_DTYPES and 1 + len(branches) <= 13)) and act[0] in (_lib.ACT_NONE, _lib.ACT_RELU, _lib.ACT_HARDTANH) and Cout % 8 == 0 and (weight.requires_grad or x_needs_grad or gamma.requires_grad):  # (a layer with residual inputs: its backward then needs none of them to re-derive the pre-activation, functional 'reduce_many')
			gate = torch.empty(B * Tout * Cout // 8, dtype = torch.uint8, device = dev)
		# planes_out (models.ConvBn1d._planes_out: the one reader of this output is a split conv): the activation pass writes the output's three
		# 16-bit planes and nothing else -- no fp32 z, no split pass in the consumer (4 + 4 + 6 bytes per element of traffic become 6)
		planes_out = bool(cfg.get('planes_out')) and cfg.get('split') is not None and dt == torch.float32 and Cout % 8 == 0 and act[0] != _lib.ACT_LEAKY_RELU
		z = ops.bn_act(y, bnp[2], bnp[3], act, xlen = xl, res = res_y, rscale = [None if p is None else p[2] for p in res_bnp], rshift = [None if p is None else p[3] for p in res_bnp], dropout_p = p_drop, seed = seed, offset = offset, gate = gate, step_key = skey, planes = cfg['split'] if planes_out else None)
		if planes_out:
			z = _planes_placeholder(z, B, Cout, Tout)
		ctx.gate = gate

		ctx.cfg, ctx.n_res, ctx.drop = cfg, n_res, (p_drop, seed, offset, skey)
		ctx.params = (weight, gamma, beta) + tuple(flat_res[5 * r + k] for r in range(n_res) for k in range(1, 5))
		ctx.x_needs_grad = x_needs_grad
		ctx.save_for_backward(x, y, bnp, xl, *res_x, *[t for t in res_y], *[p for p in res_bnp if p is not None])
		ctx.res_has_bn = [p is not None for p in res_bnp]
		ctx.res_split = res_split
		ctx.bwd_link = None
		if FUSE_BWD and cfg.get('fuse_bwd') and n_res == 0 and dt in ops.HALF_DTYPES and Cout % 8 == 0:
			ctx.bwd_link = dict(y = y, bnp = bnp, act = act, drop = (p_drop, seed, offset, skey), xl = xl, sums = _bwd_sums_buffer(bn, Cout, dev, B, Tout), dz = None, gate = gate)
			setattr(z, _LINK_ATTR, ctx.bwd_link)
		# GRAD_ACC: a 16-bit output that later blocks may tap as a residual input carries a gradient accumulator.  The tapping blocks' grouped
		# input-gradient launches write / add into it (and hand autograd None); this layer's backward adds it to the gradient autograd
		# delivers (the main path's): one explicit add per tapped output instead of autograd's pairwise one per branch.
		ctx.gacc = None
		if GROUP_RES and dt in ops.HALF_DTYPES and cfg.get('tappable'):
			ctx.gacc = dict(buf = None)
			setattr(z, _GACC_ATTR, ctx.gacc)
		return z

	@staticmethod
	def backward(ctx, dz):
		cfg, n_res = ctx.cfg, ctx.n_res
		spec, dt, act = cfg['spec'], cfg['compute_dtype'], cfg['act']
		saved = ctx.saved_tensors
		x, y, bnp, xl = saved[:4]
		res_x = list(saved[4:4 + n_res])
		res_y = list(saved[4 + n_res:4 + 2 * n_res])
		it = iter(saved[4 + 2 * n_res:])
		res_bnp = [next(it) if has else None for has in ctx.res_has_bn]
		weight, gamma, beta = ctx.params[:3]
		p_drop, seed, offset, skey = ctx.drop
		B, Cout, Tout = y.shape
		dev = y.device
		dz = ops.as_cl(dz, dt)
		hi = ctx.split is not None and bool(cfg.get('split_hi_bwd'))  # split forward, ONE 16-bit product per gradient (compute types 'bf16x3f' / 'f16x3f')
		if ctx.gacc is not None and ctx.gacc['buf'] is not None:  # the input gradients of the residual branches that tapped this output
			acc, ctx.gacc['buf'] = ctx.gacc['buf'], None
			dz = ops.add16(dz, acc, out = acc)
		link, fused_sums = ctx.bwd_link, None
		if link is not None:
			if link['dz'] is not None and link['dz'].data_ptr() == dz.data_ptr() and link['dz'].shape == dz.shape:
				fused_sums = link['sums']
			link['dz'] = None

		if n_res == 0 and fused_sums is not None:
			# pass 1 already ran inside the dgrad launch that produced dz: only the per-channel finalize is left
			coef = torch.empty(3 * Cout, dtype = torch.float32, device = dev)
			finalize = lambda outs, acc: ops.bn_bwd_finalize(fused_sums, gamma, bnp[0], bnp[1], B * Tout, coef = coef, dgamma = outs[0], dbeta = outs[1], accumulate = acc)
			if gamma.requires_grad or beta.requires_grad:
				dgamma, dbeta = _deliver([gamma, beta], finalize)
			else:
				finalize([None, None], False)
				dgamma = dbeta = None
			dy = ops.bn_act_bwd_apply(dz, y, coef, True, bnp[2], bnp[3], act, xlen = xl, dropout_p = p_drop, seed = seed, offset = offset, gate = ctx.gate, step_key = skey, planes = ctx.split, hi_only = hi)
			g = rsum_of = None
		elif n_res == 0:
			# no residuals: pass 1 only reduces (g is not materialised), its finalize kernel emits dgamma / dbeta and the three
			# per-channel coefficients, pass 2 recomputes g from dz on the fly: dy = A*g + Bc*y + D
			coef = torch.empty(3 * Cout, dtype = torch.float32, device = dev)
			reduce = lambda outs, acc: ops.bn_act_bwd_reduce(dz, y, bnp[2], bnp[3], bnp[0], bnp[1], act, xlen = xl, dropout_p = p_drop, seed = seed, offset = offset, write_g = False, gamma = gamma, coef = coef, dgamma = outs[0], dbeta = outs[1], accumulate = acc, gate = ctx.gate, step_key = skey)
			if gamma.requires_grad or beta.requires_grad:
				dgamma, dbeta = _deliver([gamma, beta], reduce)
			else:
				reduce([None, None], False)
				dgamma = dbeta = None
			dy = ops.bn_act_bwd_apply(dz, y, coef, True, bnp[2], bnp[3], act, xlen = xl, dropout_p = p_drop, seed = seed, offset = offset, gate = ctx.gate, step_key = skey, planes = ctx.split, hi_only = hi)  # (a split-operand conv: dy straight into its planes)
			g = rsum_of = None
		else:
			bn_idx = [r for r in range(n_res) if res_bnp[r] is not None]
			grouped_bn = None
			if ctx.gate is not None and GROUP_RES and dt in ops.HALF_DTYPES and 1 + len(bn_idx) <= 13:
				# the whole of pass 1 in ONE sweep from the stored gates: g written once, sum g / sum g xhat of the main batch norm and of every
				# branch's, then their finalize; pass 2 = one grouped apply
				sets = [(gamma, beta, bnp, y)] + [(ctx.params[3 + 4 * r + 2], ctx.params[3 + 4 * r + 3], res_bnp[r], res_y[r]) for r in bn_idx]
				coefs = torch.empty(len(sets), 3 * Cout, dtype = torch.float32, device = dev)
				holder = {}
				def many(outs, accs):
					holder['g'] = ops.bn_bwd_reduce_many(dz, ctx.gate, p_drop, [t[3] for t in sets], [t[2][0] for t in sets], [t[2][1] for t in sets], [t[0] for t in sets], [coefs[i] for i in range(len(sets))], [o[0] for o in outs], [o[1] for o in outs], accs)
				dgb = _deliver_many([[gm, bt] for gm, bt, _, _ in sets], many)
				g = holder['g']
				dys = ops.bn_bwd_apply_grouped(g, [t[3] for t in sets], [coefs[i] for i in range(len(sets))])
				(dgamma, dbeta), dy = dgb[0], dys[0]
				grouped_bn = {r: (dgb[1 + i][0], dgb[1 + i][1], dys[1 + i]) for i, r in enumerate(bn_idx)}
				rsum_of = None
		if n_res > 0 and grouped_bn is None:
			sums = torch.empty(2 * Cout * (1 + len(bn_idx)), dtype = torch.float64, device = dev)  # written by the reduce kernels
			rsum_of = {r: sums[2 * Cout * (1 + i):2 * Cout * (2 + i)] for i, r in enumerate(bn_idx)}
			common = dict(xlen = xl, res = res_y, rscale = [None if p is None else p[2] for p in res_bnp], rshift = [None if p is None else p[3] for p in res_bnp], rmean = [None if p is None else p[0] for p in res_bnp], rinvstd = [None if p is None else p[1] for p in res_bnp], dropout_p = p_drop, seed = seed, offset = offset, step_key = skey)
			# the kernel reduces the main BN plus the first two batch-normed residuals per pass; dense blocks with more take extra passes
			first = [r for r in bn_idx if r < 2]
			g = ops.bn_act_bwd_reduce(dz, y, bnp[2], bnp[3], bnp[0], bnp[1], act, rsums = [rsum_of.get(r) if r in first else None for r in range(n_res)], sums = sums[:2 * Cout], **common)
			rest = [r for r in bn_idx if r >= 2]
			while rest:
				batch, rest = rest[:2], rest[2:]
				# g is materialised by now: the extra passes reduce it against two more residual branches each, reading g and those two
				# tensors only (identity activation on g itself) instead of re-deriving g from dz and ALL residual inputs
				ops.bn_act_bwd_reduce(g, g, None, None, None, None, (_lib.ACT_NONE, 0.0, 0.0), res = [res_y[r] for r in batch], rscale = [None] * len(batch), rshift = [None] * len(batch), rmean = [common['rmean'][r] for r in batch], rinvstd = [common['rinvstd'][r] for r in batch], rsums = [rsum_of[r] for r in batch], write_g = False)
			grouped_bn = GROUP_RES and dt in ops.HALF_DTYPES and len(bn_idx) >= 1 and len(bn_idx) <= 12
			if grouped_bn:
				# pass 2 of the main batch norm and of every branch's: ONE finalize launch (sums -> coefficients, dgamma / dbeta) and ONE apply
				# launch that reads g once and writes all the dy_i
				coefs = torch.empty(1 + len(bn_idx), 3 * Cout, dtype = torch.float32, device = dev)
				sets = [(gamma, beta, bnp, sums[:2 * Cout])] + [(ctx.params[3 + 4 * r + 2], ctx.params[3 + 4 * r + 3], res_bnp[r], rsum_of[r]) for r in bn_idx]
				dgb = _deliver_many([[gm, bt] for gm, bt, _, _ in sets], lambda outs, accs: ops.bn_bwd_finalize_grouped([t[3] for t in sets], [t[0] for t in sets], [t[2][0] for t in sets], [t[2][1] for t in sets], B * Tout,
					[coefs[i] for i in range(len(sets))], [o[0] for o in outs], [o[1] for o in outs], accs))
				dys = ops.bn_bwd_apply_grouped(g, [y] + [res_y[r] for r in bn_idx], [coefs[i] for i in range(len(sets))])
				(dgamma, dbeta), dy = dgb[0], dys[0]
				grouped_bn = {r: (dgb[1 + i][0], dgb[1 + i][1], dys[1 + i]) for i, r in enumerate(bn_idx)}
			else:
				dgamma, dbeta, dy = _bn_backward_from_g(g, y, gamma, beta, bnp, sums[:2 * Cout], B * Tout)

		if n_res == 0:
			grouped_bn = None
		arena_mode = getattr(weight, '_convasr_grad', None) is not None
		if ctx.split is not None and dy.dtype == torch.float32:  # (the residual forms of the BN backward deliver fp32; the residual-free one wrote the planes itself)
			dy = ops.as_cl(dy, ctx.split) if hi else ops.split3(dy, ctx.split, ops.SPLIT_GRAD)
		if hi and ctx.fold is not None:
			wg = lambda: _deliver([weight], lambda outs, acc: Fold2.wgrad_hi(x, dy, weight, spec, ctx.fold[0], ctx.fold[1], outs[0], acc))
		elif hi:
			# the hi plane of the forward's saved planes, read in place (frames 3 Cin elements apart), against the 16-bit dy: x_hi dy_hi alone
			wg = lambda: _deliver([weight], lambda outs, acc: ops.conv1d_wgrad_hi(x, dy, Cout, spec.K, spec.dilation, spec.padding, outs[0], accumulate = acc))
		elif ctx.split is not None and ctx.fold is not None:
			wg = lambda: _deliver([weight], lambda outs, acc: Fold2.wgrad_split(x, dy, weight, spec, ctx.fold[0], ctx.fold[1], outs[0], acc))
		elif ctx.split is not None:
			# split-operand conv: dy as its three planes (hi, hi, lo) once, for both gradients; the weight gradient pairs plane p of x with plane p of
			# dy by reading both plane tensors as 3 T frames of C channels with the conv's dilation and padding tripled
			xf, dyf = ops.split3_frames(x), ops.split3_frames(dy)
			wg = lambda: _deliver([weight], lambda outs, acc: ops.conv1d_wgrad(xf, dyf, Cout, spec.K, 1, 3 * spec.dilation, 3 * spec.padding, outs[0], accumulate = acc, work = 2.0 * B * Tout * Cout * weight.shape[1] * spec.K, family = SPLIT_WGRAD_FAMILY))
		elif ctx.fold is not None:
			Bx, Cin, Tin = x.shape
			xv = x.as_strided((Bx, 2 * Cin, Tin // 2), (Tin * Cin, 1, 2 * Cin))
			wg = lambda: _deliver([weight], lambda outs, acc: Fold2.wgrad(xv, dy, weight, spec, ctx.fold[0], ctx.fold[1], outs[0], acc))
		else:
			wg = lambda: _deliver([weight], lambda outs, acc: ops.conv1d_wgrad(x, dy, Cout, spec.K, spec.stride, spec.dilation, spec.padding, outs[0], accumulate = acc))
		after = WGRAD_AFTER_DGRAD and arena_mode and _side_streams.get(dev) is not None
		if arena_mode and not after:
			# enqueue before dgrad: both only read dy, and the side stream can start while dgrad is still being issued
			dw, = _run_wgrad(dev, (x, dy), wg)
		dx = None
		if ctx.x_needs_grad:
			if spec.stride != 1:
				raise _lib.ConvasrHipError('conv1d dgrad with stride > 1 is not implemented (only the prologue conv is strided and its input needs no gradient)')
			dx = _dgrad(x, dy, weight, spec, dt, ctx.producer_link, split = ctx.split, hi = hi)
		if after:
			dw, = _run_wgrad(dev, (x, dy), wg)  # behind this layer's dgrad on the side stream; the next dgrad waits for it (_join_pending_wgrad)
		if not arena_mode:
			dw, = wg()

		res_grads = [None] * (5 * n_res)
		pending = []  # branches whose input gradient goes into a gradient accumulator: (r, dry), launched together below
		wg_group = []  # branches whose weight gradient goes into the gradient arena: (r, rx, dry), launched together below
		for r in range(n_res):
			rw, rb, rg, rbeta = ctx.params[3 + 4 * r:3 + 4 * r + 4]
			rx = res_x[r]
			need_rx = ctx.needs_input_grad[6 + 5 * r]
			if res_bnp[r] is None:
				res_grads[5 * r] = g if need_rx else None
				continue
			p = res_bnp[r]
			drg, drbeta, dry = grouped_bn[r] if grouped_bn else _bn_backward_from_g(g, res_y[r], rg, rbeta, p, rsum_of[r], B * Tout)
			drx = None
			if ctx.res_split[r]:
				# split branch: dry as planes once; the input gradient as a split one-tap conv, the weight gradient over the planes read as frames
				sp, cin_r = cfg['split'], rw.shape[1]
				hi_r = bool(cfg.get('split_hi_bwd'))
				dry3 = ops.as_cl(dry, sp) if hi_r else ops.split3(dry, sp, ops.SPLIT_GRAD)
				if need_rx and hi_r:
					drx = ops.conv1d(dry3, split_weight(rw, sp, 1)[1], cin_r, 1, 1, 1, 0, out_dtype = torch.float32)
				elif need_rx:
					drx = ops.conv1d(dry3, split_weight(rw, sp)[1], cin_r, 1, 1, 1, 0, out_dtype = torch.float32, work = 2.0 * B * Tout * Cout * cin_r, family = SPLIT_FAMILY)

				def res_wgrad3(outs, acc, rx3 = rx, dry3 = dry3, rb = rb, cin_r = cin_r, hi_r = hi_r):
					if hi_r:
						ops.conv1d_wgrad_hi(rx3, dry3, Cout, 1, 1, 0, outs[0], accumulate = acc)
					else:
						ops.conv1d_wgrad(ops.split3_frames(rx3), ops.split3_frames(dry3), Cout, 1, 1, 3, 0, outs[0], accumulate = acc, work = 2.0 * B * Tout * Cout * cin_r, family = SPLIT_WGRAD_FAMILY)
					if outs[1] is not None and not acc and not (outs[1] is getattr(rb, '_convasr_grad', None) and getattr(rb, '_convasr_grad_is_zero', False)):
						outs[1].zero_()  # (the bias of a conv that feeds a train-mode batch norm: an identically zero gradient, see below)
						rb._convasr_grad_is_zero = outs[1] is getattr(rb, '_convasr_grad', None)
				drw, drb = _deliver([rw, rb], res_wgrad3)
				res_grads[5 * r:5 * r + 5] = [drx, drw, drb, drg, drbeta]
				continue
			if need_rx and ctx.res_gacc[r] is not None and rx.shape[1] % 128 == 0 and Cout % 64 == 0:
				pending.append((r, dry))
			elif need_rx:
				join_prepack(dry.device)
				drx = ops.conv1d(dry, packed_weight(rw, dt, _lib.PACK_DGRAD), rx.shape[1], 1, 1, 1, 0)
			# The bias of a conv that feeds a train-mode batch norm has an identically zero gradient: dry sums to zero over (b, t) for
			# every channel (sum of g minus N times its mean, minus mean(g xhat) times sum of xhat = 0).  The reference's autograd
			# gets rounding noise around 0 from the column sum of dry; here the entry is set to exact zero and the pass is skipped.
			arena = getattr(rw, '_convasr_grad', None) is not None and (rb is None or getattr(rb, '_convasr_grad', None) is not None)
			if GROUP_RES and arena and rw.requires_grad and dt in ops.HALF_DTYPES and rx.shape[1] % 128 == 0 and Cout % 128 == 0 and (rb is None or rb.requires_grad):
				wg_group.append((r, rx, dry))  # all such branches' weight gradients in one dispatch, below
				res_grads[5 * r:5 * r + 5] = [drx, None, None, drg, drbeta]
				continue
			def res_wgrad(outs, acc, rx = rx, dry = dry, rb = rb):
				ops.conv1d_wgrad(rx, dry, Cout, 1, 1, 1, 0, outs[0], accumulate = acc)
				if outs[1] is not None and not acc and not (outs[1] is getattr(rb, '_convasr_grad', None) and getattr(rb, '_convasr_grad_is_zero', False)):
					outs[1].zero_()
					# the arena segment of this bias gradient is written by nobody else (zero at allocation, zero again now, sums of zeros
					# under data parallelism): later steps skip the fill -- dense blocks carry up to ten such biases
					rb._convasr_grad_is_zero = outs[1] is getattr(rb, '_convasr_grad', None)
			if RES_WGRAD_SIDE and arena:
				# gradient arenas: the weight gradient of the branch goes to the wgrad side stream like the main conv's (it only reads rx and dry
				# and writes the arena: the consumers of the arena join that stream, functional.join_side_streams / the data-parallel engine)
				drw, drb = _run_wgrad(dry.device, (rx, dry), lambda rw = rw, rb = rb, f = res_wgrad: _deliver([rw, rb], f))  # (bound now: the loop variables move on)
			else:
				drw, drb = _deliver([rw, rb], res_wgrad)
			res_grads[5 * r:5 * r + 5] = [drx, drw, drb, drg, drbeta]
		if pending:
			# the branches' input gradients, all in one dispatch, straight into the tapped outputs' gradient accumulators (first writer of a
			# step allocates and writes, later ones add): autograd gets None for these inputs
			join_prepack(dev)
			accs = [ctx.res_gacc[r] for r, _ in pending]
			fresh = [a['buf'] is None for a in accs]
			for (r, _), a, fr in zip(pending, accs, fresh):
				if fr:
					a['buf'] = ops.empty_cl(B, res_x[r].shape[1], Tout, dt, dev)
			done = ops.conv1x1_grouped([dry for _, dry in pending], [packed_weight(ctx.params[3 + 4 * r], dt, _lib.PACK_DGRAD) for r, _ in pending], [res_x[r].shape[1] for r, _ in pending], outs = [a['buf'] for a in accs], accumulate = [not fr for fr in fresh])
			if done is None:
				raise _lib.ConvasrHipError('grouped input gradient of the residual branches: a shape left the one-tap kernel\'s envelope between forward and backward')
		if wg_group:
			ps = [(ctx.params[3 + 4 * r], ctx.params[3 + 4 * r + 1]) for r, _, _ in wg_group]
			fresh = [bool(getattr(w, '_convasr_fresh', True)) for w, _ in ps]
			if any(b is not None and bool(getattr(b, '_convasr_fresh', True)) != fr for (_, b), fr in zip(ps, fresh)):
				raise _lib.ConvasrHipError('gradient arenas of one residual branch are out of step (mixed fresh / accumulated state)')

			def run():
				# a fresh bias gradient is zeroed by the combine kernel (no flag to go stale: every step rewrites it); an accumulated one gets += 0
				ok = ops.wgrad1x1_grouped([rx for _, rx, _ in wg_group], [dry for _, _, dry in wg_group], [w._convasr_grad for w, _ in ps], zeros = [b._convasr_grad if (b is not None and fr) else None for (_, b), fr in zip(ps, fresh)], accumulate = [not fr for fr in fresh])
				if not ok:
					raise _lib.ConvasrHipError('grouped weight gradient of the residual branches: a shape outside the kernel\'s envelope')
			if RES_WGRAD_SIDE:
				_run_wgrad(dev, tuple(t for _, rx, dry in wg_group for t in (rx, dry)), run)
			else:
				run()
			for w, b in ps:
				for p_ in (w, b):
					if p_ is not None:
						p_._convasr_fresh = False
						p_._convasr_grad_is_zero = False
						hook = getattr(p_, '_convasr_ready', None)
						if hook is not None:
							hook(p_)
		return (None, dx, dw, dgamma, dbeta, None, *res_grads)


def _momentum(bn):
	if bn.momentum is None:  # cumulative moving average (reset_bn_running_stats_, models.py:731)
		if CAPTURING[0]:
			raise _lib.ConvasrHipError('a batch norm with momentum = None (cumulative average: its factor is read from the device every step) cannot be captured into a step graph')
		return 1.0 / float(int(bn.num_batches_tracked.item()) + 1)
	return bn.momentum


def _stats_buffer(bn, C, dev, B, Tout, slot = '_convasr_stats'):
	"""Persistent per-BatchNorm partial-sum buffer (ops.ConvStats) for the conv epilogue's statistics; `slot` picks the forward
	one or the one the fused backward epilogue fills.  Nothing to zero: every launch overwrites the rows it reports."""
	if CAPTURING[0]:
		return ops.ConvStats(C, B, Tout, dev)  # allocated inside the capture (the graph's own pool): a buffer cached on the module could be replaced -- and freed -- by a later, larger batch
	st = getattr(bn, slot, None)
	if st is None or not st.fits(C, B, Tout, dev):
		st = ops.ConvStats(C, B, Tout, dev)
		setattr(bn, slot, st)
	return st


HEAD_PAD = 128  # channels the gradient of a narrow 1x1 head is padded to in backward


class _HeadPad:
	"""Backward of a 1x1 conv head with a ragged class count (the 38-class decoder, models.py:26) through the LDS-DMA kernels: the
	head's output gradient is converted to the 16-bit storage type into a zero-padded (B, 128, T) tensor, the head's weight is transposed into a
	zero-padded [Cin][128] dgrad operand, and dgrad / wgrad run as 128-channel problems (the padding contributes exact zeros;
	rows 38.. of the padded weight gradient are dropped).  The general register-staged kernels took 61 + 109 us per step for the
	3.75 GFLOP involved; these are memory-bound launches of ~30 / ~45 us, and the dgrad can carry the BN-backward sums of the last
	encoder layer in its epilogue like every other dgrad."""
	_cache = {}

	@classmethod
	def dgrad_weight(cls, weight, dt):
		Cout, Cin, K = weight.shape
		ver = param_version(weight)
		ent = cls._cache.get((id(weight), dt))
		if ent is None:
			ent = cls._cache[(id(weight), dt)] = dict(w = weight, ver = None, wd = torch.zeros(1, ops.cout_pad(Cin), HEAD_PAD, dtype = dt, device = weight.device))
		if ent['ver'] != ver:
			src = weight.detach()
			# wd[0][ci][co] = w[co][ci][0]: the layout kernel reads (C = co, T = ci) and writes it channels-last with a row pitch of 128
			_lib.call('convasr_convert_layout', _lib.ptr(src), _lib.F32, 0, src.stride(0), src.stride(1), _lib.ptr(ent['wd']), _lib.dtype_code(dt), 0, 1, HEAD_PAD, 1, Cout, Cin, _lib.stream_ptr())
			ent['ver'] = ver
		return ent['wd']

	@classmethod
	def split_weight(cls, weight, split, dgrad_planes = 3):
		"""(forward, dgrad) planes of a narrow one-tap head in a split-operand network, as 128-class operands: the fp32 weight is copied into the
		live rows of a zero-padded (128, Cin, 1) fp32 buffer and split like any other weight; refreshed per parameter version, in place.
		dgrad_planes = 1: the dgrad operand as the ordinary 16-bit one (a one-product backward, functional.split_weight)."""
		Cout, Cin, K = weight.shape
		ver = param_version(weight)
		ent = cls._cache.get((id(weight), 'split', split, dgrad_planes))
		if ent is None:
			ent = cls._cache[(id(weight), 'split', split, dgrad_planes)] = dict(w = weight, ver = None, wp = torch.zeros(HEAD_PAD, Cin, 1, dtype = torch.float32, device = weight.device), fwd = None, dgr = None)
		if ent['ver'] != ver:
			src, wp = weight.detach(), ent['wp']
			_lib.call('convasr_convert_layout', _lib.ptr(src), _lib.F32, 0, src.stride(0), src.stride(1), _lib.ptr(wp), _lib.F32, 0, wp.stride(0), wp.stride(1), 1, Cout, Cin, _lib.stream_ptr())
			ent['fwd'], ent['dgr'] = ops.pack_weight_split3(wp, split, out = (ent['fwd'], ent['dgr']), dgrad_planes = dgrad_planes)
			ent['ver'] = ver
		return ent['fwd'], ent['dgr']

	_pad_bufs = {}

	@classmethod
	def pad_grad(cls, dy, dt):
		B, Cout, T = dy.shape
		# the padded buffer is kept per (shape, type, stream): channels >= Cout are zero from its allocation on and never written, so a step
		# only rewrites the Cout live channels (no fill launch); its readers of the previous step are ordered before this write on the same
		# stream (train_step joins the weight-gradient side stream before the optimizer)
		key = (B, T, Cout, dt, dy.device, torch.cuda.current_stream(dy.device).cuda_stream)
		out = None if CAPTURING[0] else cls._pad_bufs.get(key)
		if CAPTURING[0]:
			out = ops.zeros_cl(B, HEAD_PAD, T, dt, dy.device)  # (inside a capture: the graph's own buffer, zero-filled at every replay -- by a fill KERNEL node: torch.zeros records no memset node, train.capture_node_kinds)
		elif out is None:
			if len(cls._pad_bufs) >= 64:
				cls._pad_bufs.clear()  # mixed-length training: one entry per padded length; bounded
			out = cls._pad_bufs[key] = ops.zeros_cl(B, HEAD_PAD, T, dt, dy.device)
		_lib.call('convasr_convert_layout', _lib.ptr(dy), _lib.dtype_code(dy.dtype), dy.stride(0), dy.stride(1), dy.stride(2), _lib.ptr(out), _lib.dtype_code(dt), out.stride(0), out.stride(1), out.stride(2), B, Cout, T, _lib.stream_ptr())
		return out


class ConvBiasFunction(torch.autograd.Function):
	"""Plain Conv1d with optional bias and fp32 output: the decoder head (models.py:26, 40)."""

	@staticmethod
	def forward(ctx, cfg, x, weight, bias):
		spec, dt = cfg['spec'], cfg['compute_dtype']
		ctx.producer_link = _take_link(x)
		x = ops.as_cl(x, dt)
		Cout, Cin, K = weight.shape
		ctx.split = None
		if cfg.get('split') is not None and dt == torch.float32 and K == 1 and spec.stride == 1 and spec.padding == 0 and Cin % 128 == 0 and Cout <= HEAD_PAD and os.environ.get('CONVASR_NO_HEAD_PAD') != '1':
			# the head of a split-operand network: the input's planes against the planes of the weight padded to 128 classes (rows >= Cout are zero
			# and the kernel stores the Cout live columns only); backward runs dgrad / wgrad as 128-class split problems like _HeadPad's 16-bit form
			ctx.split = cfg['split']
			B, _, T = x.shape
			x = ops.split3(x, ctx.split, ops.SPLIT_INPUT)
			y = ops.conv1d(x, _HeadPad.split_weight(weight, ctx.split, 1 if cfg.get('split_hi_bwd') else 3)[0], Cout, 1, 1, 1, 0, out_dtype = cfg.get('out_dtype', torch.float32), bias = bias, work = 2.0 * B * T * Cout * Cin, family = SPLIT_FAMILY)
		else:
			y = ops.conv1d(x, packed_weight(weight, dt, _lib.PACK_FWD), weight.shape[0], spec.K, spec.stride, spec.dilation, spec.padding, out_dtype = cfg.get('out_dtype', torch.float32), bias = bias)
		ctx.cfg = cfg
		ctx.params = (weight, bias)
		ctx.save_for_backward(x)
		return y

	@staticmethod
	def backward(ctx, dy):
		cfg = ctx.cfg
		spec, dt = cfg['spec'], cfg['compute_dtype']
		weight, bias = ctx.params
		x, = ctx.saved_tensors
		Cout = weight.shape[0]
		if ctx.split is not None:
			Cin = weight.shape[1]
			B, _, T = dy.shape
			dy = ops.as_cl(dy, torch.float32)
			hi = bool(cfg.get('split_hi_bwd'))  # one 16-bit product per gradient: dy rounded once into the padded 128-class buffer, x_hi read in place, w_hi
			dy3 = _HeadPad.pad_grad(dy, ctx.split) if hi else ops.split3(_HeadPad.pad_grad(dy, torch.float32), ctx.split, ops.SPLIT_GRAD)  # (B, 3 x 128, T): the padded classes are zero planes
			dx = None
			if ctx.needs_input_grad[1] and hi:
				dx = ops.conv1d(dy3, _HeadPad.split_weight(weight, ctx.split, 1)[1], Cin, 1, 1, 1, 0, out_dtype = torch.float32, work = 2.0 * B * T * Cout * Cin)
				_after_long_launch()
			elif ctx.needs_input_grad[1]:
				dx = ops.conv1d(dy3, _HeadPad.split_weight(weight, ctx.split)[1], Cin, 1, 1, 1, 0, out_dtype = torch.float32, work = 2.0 * B * T * Cout * Cin, family = SPLIT_FAMILY)
				_after_long_launch()

			def wgrad3(outs, acc):
				dwp = torch.empty(HEAD_PAD, Cin, 1, dtype = torch.float32, device = dy.device)
				if hi:
					ops.conv1d_wgrad_hi(x, dy3, HEAD_PAD, 1, 1, 0, dwp, work = 2.0 * B * T * Cout * Cin)
				else:
					ops.conv1d_wgrad(ops.split3_frames(x), ops.split3_frames(dy3), HEAD_PAD, 1, 1, 3, 0, dwp, work = 2.0 * B * T * Cout * Cin, family = SPLIT_WGRAD_FAMILY)
				if outs[0] is not None:
					v = dwp[:Cout]
					if acc:
						outs[0].add_(v.view(outs[0].shape))
					else:  # rows < Cout of the padded result into the gradient (arena) view, element (c, t) = (co, ci)
						_lib.call('convasr_convert_layout', _lib.ptr(v), _lib.F32, 0, v.stride(0), v.stride(1), _lib.ptr(outs[0]), _lib.F32, 0, outs[0].stride(0), outs[0].stride(1), 1, Cout, Cin, _lib.stream_ptr())
				if outs[1] is not None:
					ops.colsum(dy, outs[1], accumulate = acc)  # the bias gradient: the sum of dy itself, not of its planes
			dw, db = _deliver([weight, bias], wgrad3)
			return None, dx, dw, db
		if dt in ops.HALF_DTYPES and spec.K == 1 and spec.stride == 1 and spec.padding == 0 and Cout < HEAD_PAD and x.shape[1] % 128 == 0 and os.environ.get('CONVASR_NO_HEAD_PAD') != '1':
			dyp = _HeadPad.pad_grad(dy, dt)
			dx = _dgrad(x, dyp, weight, spec, dt, ctx.producer_link, wd = _HeadPad.dgrad_weight(weight, dt)) if ctx.needs_input_grad[1] else None

			def wgrad(outs, acc):
				dwp = torch.empty(HEAD_PAD, x.shape[1], 1, dtype = torch.float32, device = x.device)
				dbp = torch.empty(HEAD_PAD, dtype = torch.float32, device = x.device) if outs[1] is not None else None
				ops.conv1d_wgrad(x, dyp, HEAD_PAD, 1, 1, 1, 0, dwp, dbias = dbp)
				for o, v in ((outs[0], dwp[:Cout]), (outs[1], None if dbp is None else dbp[:Cout].view(Cout, 1, 1))):
					if o is None:
						continue
					if acc:
						o.add_(v.view(o.shape))
					else:  # rows < Cout of the padded result into the gradient (arena) view: our own strided copy kernel, element (c, t) = (co, ci)
						ov = o.view(Cout, -1, 1) if o.ndim == 1 else o
						_lib.call('convasr_convert_layout', _lib.ptr(v), _lib.F32, 0, v.stride(0), v.stride(1), _lib.ptr(ov), _lib.F32, 0, ov.stride(0), ov.stride(1), 1, Cout, v.shape[1], _lib.stream_ptr())
			dw, db = _deliver([weight, bias], wgrad)
			return None, dx, dw, db
		dy = ops.as_cl(dy, dt)
		dx = None
		if ctx.needs_input_grad[1]:
			if spec.stride != 1:
				raise _lib.ConvasrHipError('conv1d dgrad with stride > 1 is not implemented')
			dx = _dgrad(x, dy, weight, spec, dt, ctx.producer_link)
		dw, db = _deliver([weight, bias], lambda outs, acc: ops.conv1d_wgrad(x, dy, Cout, spec.K, spec.stride, spec.dilation, spec.padding, outs[0], dbias = outs[1], accumulate = acc))
		return None, dx, dw, db


class GroupedConvReluFunction(torch.autograd.Function):
	"""The first half of the reference's separable block (models.py:50-64): nn.Conv1d(Cin, Cout, K, groups = G) with its bias, then ReLU;
	the 1x1 conv that follows is an ordinary ConvBn1d repeat.  apply(cfg, x, weight, bias), cfg: dict(spec, groups, compute_dtype)."""

	@staticmethod
	def forward(ctx, cfg, x, weight, bias):
		spec, dt = cfg['spec'], cfg['compute_dtype']
		_take_link(x)  # (no cross-layer fusion through a grouped conv: a producer's link, if any, is dropped and that layer reduces on its own)
		x = ops.as_cl(x, dt)
		y = ops.grouped_conv1d(x, weight.detach(), None if bias is None else bias.detach(), cfg['groups'], spec.stride, spec.padding, relu = True)
		ctx.cfg, ctx.params = cfg, (weight, bias)
		ctx.save_for_backward(x, y)
		return y

	@staticmethod
	def backward(ctx, dy):
		cfg = ctx.cfg
		spec, dt, groups = cfg['spec'], cfg['compute_dtype'], cfg['groups']
		weight, bias = ctx.params
		x, y = ctx.saved_tensors
		dy = ops.as_cl(dy, dt)
		dx = ops.grouped_conv1d_dgrad(dy, y, weight.detach(), x.shape[1], x.shape[2], groups, spec.stride, spec.padding) if ctx.needs_input_grad[1] else None
		dw, db = _deliver([weight, bias], lambda outs, acc: ops.grouped_conv1d_wgrad(x, dy, y, outs[0], outs[1], groups, spec.stride, spec.padding, accumulate = acc))
		return None, dx, dw, db


class ConvBnActEvalFunction:
	"""Inference path (BN folded into the conv epilogue's scale/shift, or a fused bias after fuse_conv_bn_eval): no graph."""

	@staticmethod
	def apply(cfg, x, weight, bias, scale_shift, xlen, res_list):
		spec, dt, act = cfg['spec'], cfg['compute_dtype'], cfg['act']
		x = ops.as_cl(x, dt)
		Cout = weight.shape[0]
		xl = ops.xlen_f32(xlen, x.device) if (cfg['temporal_mask'] and xlen is not None) else None
		scale, shift = (None, None) if scale_shift is None else (scale_shift[0], scale_shift[1])
		fold = Fold2.plan(x, weight, spec, dt, False)
		# split-K for launches of a few tiles (ops.conv1d): inference only, i.e. no gradient is being recorded -- a frozen block inside a training step keeps
		# the unsplit kernel's association of the sum (its output feeds a chaotic two-step golden comparison, tests/test_training_features_gpu.py, and nothing is latency-bound there)
		sk = not torch.is_grad_enabled()
		split = cfg.get('split_eval')  # (inference on the split-operand path: JasperNet.set_compute_dtype('bf16x3', inference = True))
		if fold is not None:
			xv, Kf, Pf, Tout = fold
			conv = lambda **epilogue: ops.conv1d(xv, Fold2.packed_weight(weight, dt, spec.padding), Cout, Kf, 1, 1, Pf, bias = bias, Tout = Tout, work = 2.0 * x.shape[0] * Tout * Cout * x.shape[1] * spec.K, splitk = sk, **epilogue)
		elif split_applies(split, dt, spec, x.shape[1], Cout):
			# fp32 activations, the conv as hi*hi + hi*lo + lo*hi on the 16-bit matrix pipe (csrc/split3.hip): bias / folded BN / activation / mask in
			# the conv's own epilogue as on every other path; the weights' planes are packed once per parameter version
			x3, wp3 = ops.split3(x, split, ops.SPLIT_INPUT), split_weight(weight, split)[0]
			conv = lambda **epilogue: ops.conv1d(x3, wp3, Cout, spec.K, 1, spec.dilation, spec.padding, out_dtype = torch.float32, bias = bias, work = 2.0 * x.shape[0] * ops.conv_out_len(x.shape[2], spec.K, 1, spec.dilation, spec.padding) * Cout * x.shape[1] * spec.K, family = SPLIT_FAMILY, splitk = sk, **epilogue)
		else:
			wp = packed_weight(weight, dt, _lib.PACK_FWD)
			conv = lambda **epilogue: ops.conv1d(x, wp, Cout, spec.K, spec.stride, spec.dilation, spec.padding, bias = bias, splitk = sk, **epilogue)  # (splitk: a launch of a few tiles -- one online request -- is cut over the input channels, ops.conv1d)
		# dropout > 0 here means a FROZEN block (JasperNet.freeze, models.py:328-339): its batch norms run on their running statistics
		# but the block is still in training mode, and the reference's ResidualActivation applies dropout by self.training (models.py:365-369)
		p_drop = float(cfg.get('dropout_p', 0.0))
		if not res_list and p_drop == 0:
			return conv(scale = scale, shift = shift, act = act, xlen = xl)
		y = conv()
		res_y, rscale, rshift = [], [], []
		for rx, rw, rb, rss in res_list:
			rx = ops.as_cl(rx, dt)
			if rw is None:
				res_y.append(rx); rscale.append(None); rshift.append(None)
			elif split_applies(split, dt, ConvSpec(1), rx.shape[1], Cout):
				res_y.append(ops.conv1d(ops.split3(rx, split, ops.SPLIT_INPUT), split_weight(rw, split)[0], Cout, 1, 1, 1, 0, out_dtype = torch.float32, bias = rb, work = 2.0 * rx.shape[0] * rx.shape[2] * Cout * rx.shape[1], family = SPLIT_FAMILY, splitk = sk))
				rscale.append(None if rss is None else rss[0]); rshift.append(None if rss is None else rss[1])
			else:
				res_y.append(ops.conv1d(rx, packed_weight(rw, dt, _lib.PACK_FWD), Cout, 1, 1, 1, 0, bias = rb, splitk = sk))
				rscale.append(None if rss is None else rss[0]); rshift.append(None if rss is None else rss[1])
		# rscale None (identity residual, or a residual conv already fused with its BN) means "add as is"
		seed, offset = _DropoutState.next(y.numel()) if p_drop > 0 else (0, 0)
		return ops.bn_act(y, scale, shift, act, xlen = xl, res = res_y, rscale = rscale, rshift = rshift, dropout_p = p_drop, seed = seed, offset = offset, step_key = _DropoutState.key(y.device) if p_drop > 0 else None)


class ConvBnActFrozenStatsFunction(torch.autograd.Function):
	"""One repeat of ConvBn1d whose batch norms normalise with their RUNNING statistics (`bn.eval()`: a block frozen by JasperNet.freeze
	whose input still carries a gradient, or a fine-tuning recipe that freezes the statistics only) while gradients are wanted --
	for the conv weights, gamma / beta, the input, the residual branches, or any subset.  Eval-mode batch norm is a per-channel affine map
	u = y * scale + shift with scale = gamma / sqrt(running_var + eps), so backward is g = dz * act'(.) * dropout * mask, dy = g * scale,
	dgamma = sum g * (y - running_mean) / sqrt(running_var + eps), dbeta = sum g: the training path's kernels with the batch-statistics
	terms of the coefficients set to zero (the reference gets the same from autograd through F.batch_norm(training = False)).

	apply(cfg, x, weight, bias, gamma, beta, xlen, *flat_res) with flat_res per residual (res_x, res_weight, res_bias, res_gamma, res_beta),
	the last four None for an identity residual -- ConvBnActFunction's argument list plus the conv bias (a fused conv carries one).
	cfg['bn'] / cfg['res_bn'][r]: the BatchNorm1d modules (running statistics, eps), or None / nn.Identity where fuse_conv_bn_eval removed them."""

	@staticmethod
	def _affine(bn, gamma, beta):
		if bn is None or not isinstance(bn, torch.nn.modules.batchnorm._BatchNorm):
			return None, None, None, None
		ss = ops.bn_eval_scale_shift(gamma, beta, bn.running_mean, bn.running_var, bn.eps)
		return ss[0], ss[1], bn.running_mean.detach().float(), torch.rsqrt(bn.running_var.detach().float() + bn.eps)

	@staticmethod
	def forward(ctx, cfg, x, weight, bias, gamma, beta, xlen, *flat_res):
		spec, dt, act = cfg['spec'], cfg['compute_dtype'], cfg['act']
		ctx.producer_link = _take_link(x)
		x = ops.as_cl(x, dt)
		dev = x.device
		Cout = weight.shape[0]
		xl = ops.xlen_f32(xlen, dev) if (cfg['temporal_mask'] and xlen is not None) else None
		y = ops.conv1d(x, packed_weight(weight, dt, _lib.PACK_FWD), Cout, spec.K, spec.stride, spec.dilation, spec.padding, bias = bias)
		aff = ConvBnActFrozenStatsFunction._affine(cfg['bn'], gamma, beta)
		n_res = len(flat_res) // 5
		res_x, res_y, res_aff = [], [], []
		for r in range(n_res):
			rx, rw, rb, rg, rbeta = flat_res[5 * r:5 * r + 5]
			rx = ops.as_cl(rx, dt)
			res_x.append(rx)
			if rw is None:
				res_y.append(rx); res_aff.append((None, None, None, None))
			else:
				res_y.append(ops.conv1d(rx, packed_weight(rw, dt, _lib.PACK_FWD), Cout, 1, 1, 1, 0, bias = rb))
				res_aff.append(ConvBnActFrozenStatsFunction._affine(cfg['res_bn'][r], rg, rbeta))
		p_drop = cfg['dropout_p']
		seed, offset = _DropoutState.next(y.numel()) if p_drop > 0 else (0, 0)
		skey = _DropoutState.key(dev) if p_drop > 0 else None
		z = ops.bn_act(y, aff[0], aff[1], act, xlen = xl, res = res_y, rscale = [a[0] for a in res_aff], rshift = [a[1] for a in res_aff], dropout_p = p_drop, seed = seed, offset = offset, step_key = skey)
		ctx.cfg, ctx.n_res, ctx.drop = cfg, n_res, (p_drop, seed, offset, skey)
		ctx.params = (weight, bias, gamma, beta) + tuple(flat_res[5 * r + k] for r in range(n_res) for k in range(1, 5))
		ctx.aff, ctx.res_aff = aff, res_aff
		ctx.x_needs_grad = x.requires_grad or ctx.needs_input_grad[1]
		ctx.save_for_backward(x, y, xl, *res_x, *res_y)
		return z

	@staticmethod
	def _coef(scale, C, dev):
		coef = torch.zeros(3 * C, dtype = torch.float32, device = dev)  # dy = A * g + Bc * y + D with Bc = D = 0: no batch-statistics terms
		coef[:C] = 1.0 if scale is None else scale
		return coef

	@staticmethod
	def backward(ctx, dz):
		cfg, n_res = ctx.cfg, ctx.n_res
		spec, dt, act = cfg['spec'], cfg['compute_dtype'], cfg['act']
		saved = ctx.saved_tensors
		x, y, xl = saved[:3]
		res_x, res_y = list(saved[3:3 + n_res]), list(saved[3 + n_res:3 + 2 * n_res])
		weight, bias, gamma, beta = ctx.params[:4]
		scale, shift, rmean, rinv = ctx.aff
		p_drop, seed, offset, skey = ctx.drop
		B, Cout, Tout = y.shape
		dev = y.device
		dz = ops.as_cl(dz, dt)
		live = lambda p: p is not None and p.requires_grad
		common = dict(xlen = xl, res = res_y, rscale = [a[0] for a in ctx.res_aff], rshift = [a[1] for a in ctx.res_aff], dropout_p = p_drop, seed = seed, offset = offset, step_key = skey)
		# g = dz * act' * dropout * mask, materialised (the residual branches and the weight gradients read it); with trainable gamma / beta the
		# same pass also sums g and g * (y - running_mean) * rinv per channel.  (The finalize kernel's coefficients are not used: they carry
		# the batch-statistics terms of a TRAINING batch norm.)
		dgamma = dbeta = None
		if scale is not None and (live(gamma) or live(beta)):
			sums = torch.empty(2 * Cout, dtype = torch.float64, device = dev)
			scratch = torch.empty(3 * Cout, dtype = torch.float32, device = dev)
			holder = {}
			def reduce(outs, acc):
				holder['g'] = ops.bn_act_bwd_reduce(dz, y, scale, shift, rmean, rinv, act, sums = sums, write_g = True, gamma = gamma, coef = scratch, dgamma = outs[0], dbeta = outs[1], accumulate = acc, **common)
			dgamma, dbeta = _deliver([gamma, beta], reduce)
			g = holder['g']
		else:
			g = ops.bn_act_bwd_reduce(dz, y, scale, shift, None, None, act, write_g = True, **common)
		dy = g if scale is None else ops.bn_act_bwd_apply(g, y, ConvBnActFrozenStatsFunction._coef(scale, Cout, dev), False)
		dw, db = _deliver([weight, bias], lambda outs, acc: ops.conv1d_wgrad(x, dy, Cout, spec.K, spec.stride, spec.dilation, spec.padding, outs[0], dbias = outs[1], accumulate = acc))
		dx = None
		if ctx.x_needs_grad:
			if spec.stride != 1:
				raise _lib.ConvasrHipError('conv1d dgrad with stride > 1 is not implemented (only the prologue conv is strided and its input needs no gradient)')
			dx = _dgrad(x, dy, weight, spec, dt, ctx.producer_link)
		res_grads = []
		for r in range(n_res):
			rw, rb, rg, rbeta = ctx.params[4 + 4 * r:4 + 4 * r + 4]
			need_rx = ctx.needs_input_grad[7 + 5 * r]
			if rw is None:
				res_grads += [g if need_rx else None, None, None, None, None]
				continue
			rscale, rshift, rrm, rri = ctx.res_aff[r]
			drg = drbeta = None
			if rscale is not None and (live(rg) or live(rbeta)):
				# sum g and sum g * xhat of this branch: the reduce kernel on (g, g) with an identity activation and ONE "residual" = the branch
				rs = torch.empty(2 * Cout, dtype = torch.float64, device = dev)
				ops.bn_act_bwd_reduce(g, g, None, None, None, None, (_lib.ACT_NONE, 0.0, 0.0), res = [res_y[r]], rscale = [None], rshift = [None], rmean = [rrm], rinvstd = [rri], rsums = [rs], write_g = False)
				def put(outs, acc, rs = rs):
					for o, v in ((outs[0], rs[Cout:]), (outs[1], rs[:Cout])):
						if o is not None:
							o.add_(v.float()) if acc else o.copy_(v.float())
				drg, drbeta = _deliver([rg, rbeta], put)
			dry = g if rscale is None else ops.bn_act_bwd_apply(g, res_y[r], ConvBnActFrozenStatsFunction._coef(rscale, Cout, dev), False)
			join_prepack(dev)
			drx = ops.conv1d(dry, packed_weight(rw, dt, _lib.PACK_DGRAD), res_x[r].shape[1], 1, 1, 1, 0) if need_rx else None
			if rb is not None:
				rb._convasr_grad_is_zero = False  # a real bias gradient lands in the arena segment: the train-mode path (ConvBnActFunction) must zero it again
			drw, drb = _deliver([rw, rb], lambda outs, acc, rx = res_x[r], dry = dry: ops.conv1d_wgrad(rx, dry, Cout, 1, 1, 1, 0, outs[0], dbias = outs[1], accumulate = acc))
			res_grads += [drx, drw, drb, drg, drbeta]
		return (None, dx, dw, db, dgamma, dbeta, None, *res_grads)


class LogSoftmaxFunction(torch.autograd.Function):
	"""F.log_softmax(logits, dim=1).float() (models.py:316) on channels-last fp32 logits."""

	@staticmethod
	def forward(ctx, logits):
		lp = ops.log_softmax(ops.as_cl(logits, torch.float32))
		ctx.save_for_backward(lp)
		return lp

	@staticmethod
	def backward(ctx, g):
		lp, = ctx.saved_tensors
		return ops.log_softmax_bwd(g, lp)


class CtcLossFunction(torch.autograd.Function):
	"""F.ctc_loss(lp.permute(2,0,1), y, olen, ylen, blank, reduction='none') (models.py:323): the alpha-beta kernels
	produce the per-utterance NLL and d nll / d log_probs in one pass; backward is a per-utterance scaling.
	norm (optional int64 (B,)): the result is nll / norm, the "/ ylen[:, 0]" of models.py:323; its backward rides in the same
	scaling pass instead of a chain of two-element autograd kernels."""

	@staticmethod
	def forward(ctx, log_probs, targets, olen, ylen, blank, norm = None):
		lp = ops.as_cl(log_probs, torch.float32)
		need = log_probs.requires_grad
		nll, grad = ops.ctc_loss(lp, targets, olen, ylen, blank, need_grad = need)
		ctx.norm = norm
		if need:
			ctx.save_for_backward(grad)
		if norm is None:
			return nll
		if norm.dtype == torch.int64 and norm.ndim == 1 and norm.device == nll.device:
			return ops.scale_rows(nll, None, norm)  # nll / norm in one small launch of our own (an ATen true-divide of fp32 by int64 otherwise)
		return nll / norm

	@staticmethod
	def backward(ctx, g):
		grad, = ctx.saved_tensors
		return ops.scale_rows(grad, g, ctx.norm), None, None, None, None, None


def ctc_loss(log_probs, targets, olen, ylen, blank, norm = None):
	return CtcLossFunction.apply(log_probs, targets, olen, ylen, blank, norm)
