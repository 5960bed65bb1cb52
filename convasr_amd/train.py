"""Training-step plumbing for the hot path (reference: train.py:745-783).

FlatParameters re-homes every trainable parameter into ONE contiguous fp32 buffer with a matching gradient buffer, so that
gradient clipping is one reduction launch, the optimizer one elementwise launch, and data-parallel all-reduce a handful of
large contiguous RCCL calls (288 GB of HBM makes the duplicate-free arena trivially affordable).  Backward kernels write
gradients straight into the arena (convasr_amd.functional._deliver).

Conv weights live in the arena TAP-MAJOR ([K][Cout][Cin], include/convasr_hip.h CONVASR_W_KMAJOR): the parameter stays a
(Cout, Cin, K) tensor -- state_dict(), load_state_dict() and checkpoints see the reference's shapes -- but it is a strided view.
That is the element order of the packed MFMA operand and of the weight-gradient slabs, so (a) the optimizer kernel also writes a
bf16 mirror of the arena whose conv segments ARE the packed forward weights (no packing launches per step), and (b) the split-K
combine of the weight gradient is a streaming sum instead of a transpose."""
import os

import torch

from . import ops, _lib
from . import models as M
from . import functional as Fn


class FlatParameters:
	ALIGN = 64  # elements: every parameter starts on a 256-byte boundary

	def __init__(self, model):
		params = [p for p in model.parameters() if p.requires_grad]
		if not params:
			raise ValueError('model has no trainable parameters')
		dev = params[0].device
		self.params, self.offsets = params, []
		off = 0
		for p in params:
			if p.dtype != torch.float32 or p.device != dev:
				raise ValueError('FlatParameters needs fp32 parameters on one device')
			self.offsets.append(off)
			off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
		self.numel = off
		self.data = torch.zeros(off, dtype = torch.float32, device = dev)
		self.grad = torch.zeros(off, dtype = torch.float32, device = dev)
		self.data16 = None  # 16-bit (bf16 or fp16: the model's compute dtype) mirror of `data`, allocated on first use (mirror()); written by the fused optimizer kernels
		self._mirror_ver = {}  # id(param) -> version tuple (functional.packed_weight) at which the mirror segment equals bf16(param)
		self._as_param = []  # per parameter: flat arena-shaped tensor -> view in the parameter's logical (reference) shape
		for p, o in zip(params, self.offsets):
			n = p.numel()
			kmajor = p.ndim == 3 and p.shape[2] > 1 and (p.shape[0] * p.shape[1]) % 8 == 0 and dev.type == 'cuda'
			as_param = (lambda flat, p = p, o = o, n = n: flat[o:o + n].view(p.shape[2], p.shape[0], p.shape[1]).permute(1, 2, 0)) if kmajor else (lambda flat, p = p, o = o, n = n: flat[o:o + n].view(p.shape))  # (o, n bound now: the views are rebuilt later for optimizer state)
			self._as_param.append(as_param)
			view = as_param(self.data)
			view.copy_(p.data)
			p.data = view
			p._convasr_grad = as_param(self.grad)
			p._convasr_fresh = True
			# (a one-tap conv weight, (Cout, Cin, 1) contiguous, IS its own tap-major form: its mirror segment serves as the packed forward
			# operand just the same -- the dense-residual models carry ~55 such 1x1 convs, one packing launch each per step otherwise)
			one_tap = p.ndim == 3 and p.shape[2] == 1 and (p.shape[0] * p.shape[1]) % 8 == 0 and dev.type == 'cuda'
			p._convasr_arena = (self, o) if (kmajor or one_tap) else None
			p.grad = p._convasr_grad
		self._by_id = {id(p): p for p in params}
		self.clip = None  # (sumsq double buffer, max_norm) set by clip_grad_norm_
		self.grad_scale = 1.0  # pending scale of .grad (the data-parallel engine's 1 / world_size), consumed by the next optimizer step
		self._sumsq = torch.zeros(1, dtype = torch.float64, device = dev)
		self.loss_scaler = None  # a LossScaler (fp16 training): .grad then holds gradients of the SCALED loss until the optimizer step unscales them

	def mirror(self, dtype = torch.bfloat16):
		"""The arena's 16-bit mirror in `dtype` (one at a time: asking for the other type re-allocates it and drops what was current)."""
		if self.data16 is None or self.data16.dtype != dtype:
			assert dtype in ops.HALF_DTYPES
			self.data16 = torch.zeros(self.numel, dtype = dtype, device = self.data.device)
			self._mirror_ver = {}
		return self.data16

	def mirror_carried_over(self, run):
		"""Run one fused optimizer launch `run(p16)` that rewrites the 16-bit mirror together with the parameters (or leaves both
		untouched when the device-side gate skips the step): segments that were current before it are current after it."""
		was = {k for k, v in self._mirror_ver.items() if v == Fn.param_version(self._by_id[k])} if self._mirror_ver else set()
		run(self.data16)
		Fn.bump_param_epoch()  # packed compute copies other than the mirror are stale now
		self._mirror_ver = {k: Fn.param_version(self._by_id[k]) for k in was} if self.data16 is not None else {}

	def param_views(self, flat_tensor):
		"""Per-parameter views of an arena-shaped tensor (optimizer state) in the parameters' logical shapes -- the reference's
		(Cout, Cin, K) for conv weights, whatever the arena's element order is."""
		assert flat_tensor.numel() == self.numel
		return [f(flat_tensor) for f in self._as_param]

	def export_state(self, flat_tensor):
		"""Arena-shaped optimizer state -> list of contiguous per-parameter tensors in reference shapes (layout-independent checkpoints)."""
		return [v.contiguous().clone() for v in self.param_views(flat_tensor)]

	def import_state(self, flat_tensor, per_param, what):
		if isinstance(per_param, torch.Tensor):
			raise ValueError(f'{what}: a flat arena-ordered tensor (optimizer state written before format 2) carries no layout tag and cannot be loaded safely: conv segments may be tap-major or reference-ordered')
		views = self.param_views(flat_tensor)
		if len(per_param) != len(views):
			raise ValueError(f'{what}: {len(per_param)} tensors for {len(views)} parameters')
		for v, t in zip(views, per_param):
			if tuple(v.shape) != tuple(t.shape):
				raise ValueError(f'{what}: shape {tuple(t.shape)} does not match parameter shape {tuple(v.shape)}')
			v.copy_(t)

	def zero_grad(self):
		"""No memset: the next backward overwrites (first write per parameter has accumulate = False)."""
		for p in self.params:
			p._convasr_fresh = True
			p.grad = p._convasr_grad
		self.clip = None

	def grads_written_externally(self):
		"""Call after anything other than this package's backward kernels wrote into `.grad` / `p._convasr_grad` (an optimizer that keeps its
		clipped gradients there, a test, a hand-rolled gradient edit): drops the "this segment is known to be zero" notes the train-mode
		residual branches keep for their conv biases (functional.ConvBnActFunction), so that the next backward zeroes them again."""
		for p in self.params:
			if getattr(p, '_convasr_grad_is_zero', False):
				p._convasr_grad_is_zero = False

	def finalize_grads(self):
		"""Parameters that received no gradient this step count as zero."""
		for p in self.params:
			if p._convasr_fresh:
				p._convasr_grad.zero_()

	def clip_grad_norm_(self, max_norm):
		"""torch.nn.utils.clip_grad_norm_(params, max_norm) (train.py:777): one reduction launch now; the scaling itself is
		folded into the optimizer kernel.  Returns the total norm as a 0-d device tensor (no host sync)."""
		self.finalize_grads()
		norm = torch.empty(1, dtype = torch.float32, device = self.grad.device)
		ops.sumsq(self.grad, self._sumsq, norm_out = norm, norm_scale = self.grad_scale, loss_scaler = None if self.loss_scaler is None else self.loss_scaler.current)
		self.clip = (self._sumsq, float(max_norm))
		return norm[0]


class LossScaler:
	"""Dynamic loss scaling for fp16 training: the role apex.amp's LossScaler plays behind `apex.amp.initialize(opt_level)`
	(models.py:744-762) and `with apex.amp.scale_loss(loss, optimizer)` (train.py:770-772), with apex's defaults (initial scale 2^16,
	x 2 after 2000 clean steps, / 2 and the step skipped on overflow, ceiling 2^24).  The state lives on the device
	(include/convasr_hip.h, CONVASR_LOSS_SCALER_FLOATS) in two buffers: every optimizer step reads `current` and writes the other one,
	advance() swaps them -- the host never waits for the overflow verdict.  loss_scale = a number: static scale (apex's
	loss_scale = 128.0 form), no overflow check."""

	def __init__(self, device, loss_scale = 'dynamic', init_scale = 2.0 ** 16, scale_factor = 2.0, scale_window = 2000, min_loss_scale = None, max_loss_scale = 2.0 ** 24):
		dynamic = loss_scale in (None, 'dynamic')
		state = [float(init_scale) if dynamic else float(loss_scale), 0.0, 0.0, float(scale_window) if dynamic else 0.0, float(min_loss_scale or 0.0), float(max_loss_scale), float(scale_factor), 0.0]
		assert len(state) == _lib.LOSS_SCALER_FLOATS
		self.state = torch.tensor([state, state], dtype = torch.float32, device = device)
		self.cur = 0
		self.pinned = False  # set by the first step-graph capture: from then on the current row stays where the graphs read it (see advance)

	@property
	def current(self):
		return self.state[self.cur]

	def pair(self):
		return self.state[self.cur], self.state[1 - self.cur]

	def advance(self):
		"""After an optimizer launch wrote the next state into the other buffer: swap the two -- or, once a step graph has been captured
		(its kernels read THIS buffer at every replay; `pinned`, eager steps included: optimizers._advance), enqueue the copy that hands the
		new state back to it."""
		if Fn.capturing() or self.pinned:
			_lib.call('convasr_copy', _lib.ptr(self.state[1 - self.cur]), _lib.ptr(self.state[self.cur]), 4 * _lib.LOSS_SCALER_FLOATS, _lib.stream_ptr())
		else:
			self.cur = 1 - self.cur

	def loss_scale(self):
		"""Host read (synchronises): apex's `_amp_state.loss_scalers[0].loss_scale()`."""
		return float(self.current[0])

	def state_dict(self):
		"""apex.amp.state_dict()'s entry for one scaler (train.py:332 saves it as `amp_state_dict`)."""
		s = self.current.tolist()
		return dict(loss_scale = s[0], unskipped = int(s[1]))

	def load_state_dict(self, sd):
		cur = self.current.clone()
		cur[0], cur[1] = float(sd['loss_scale']), float(sd['unskipped'])
		self.state[self.cur].copy_(cur)


def amp_state_dict(optimizer):
	"""apex.amp.state_dict() (train.py:332): {'loss_scaler0': {...}} when the optimizer trains under a loss scaler, else {}."""
	scaler = getattr(getattr(optimizer, 'flat', None), 'loss_scaler', None)
	return {} if scaler is None else dict(loss_scaler0 = scaler.state_dict())


def amp_load_state_dict(optimizer, sd):
	"""apex.amp.load_state_dict(checkpoint['amp_state_dict']) (train.py:707-708)."""
	scaler = getattr(getattr(optimizer, 'flat', None), 'loss_scaler', None)
	if scaler is not None and sd and 'loss_scaler0' in sd:
		scaler.load_state_dict(sd['loss_scaler0'])


class SGD:
	"""torch.optim.SGD semantics (train.py:657-662) as ONE fused kernel launch over the flat arena."""

	def __init__(self, flat: FlatParameters, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3, nesterov = False, keep_clipped_grads = False):
		self.flat = flat
		self.param_groups = [dict(params = flat.params, lr = lr, momentum = momentum, weight_decay = weight_decay, nesterov = nesterov)]
		self.momentum_buffer = torch.zeros_like(flat.data) if momentum != 0 else None
		self.steps = 0
		self.keep_clipped_grads = keep_clipped_grads
		self.lr_dev = None  # 1-element fp32 device tensor the captured step graphs read the learning rate from (GraphedTrainStep keeps it equal to param_groups[0]['lr'])

	def zero_grad(self, set_to_none = False):
		self.flat.zero_grad()

	def step(self, loss_gate = None):
		"""loss_gate: optional 1-element fp32 device tensor; a non-finite value turns the launch into a no-op (device-side skip)."""
		g = self.param_groups[0]
		flat = self.flat
		if flat.clip is None:
			flat.finalize_grads()
		sumsq, max_norm = flat.clip if flat.clip is not None else (None, 0.0)
		first, grad_scale = self.steps == 0, flat.grad_scale
		if first and Fn.capturing():
			raise _lib.ConvasrHipError('SGD: the first step (which initialises the momentum buffer) cannot be the one captured into a step graph')
		scaler = flat.loss_scaler
		if scaler is not None and sumsq is None:  # the overflow check reads the gradient's sum of squares
			sumsq = ops.sumsq(flat.grad, flat._sumsq)
		flat.mirror_carried_over(lambda p16: ops.sgd_step(flat.data, flat.grad, self.momentum_buffer, flat.numel, sumsq, max_norm, g['lr'], g['momentum'], g['weight_decay'], g['nesterov'], first, grad_out = flat.grad if self.keep_clipped_grads else None, loss_gate = loss_gate, grad_scale = grad_scale, p16 = p16, scaler = None if scaler is None else scaler.pair(), lr_dev = self.lr_dev if Fn.capturing() else None))
		if scaler is not None:
			scaler.advance()
		if self.keep_clipped_grads:
			flat.grads_written_externally()  # (zero times a NaN clip factor is not zero)
		self.steps += 1
		flat.clip, flat.grad_scale = None, 1.0

	def state_dict(self):
		"""format 2: momentum per parameter, in the reference's shapes (independent of the arena's tap-major element order)."""
		return dict(format = 2, steps = self.steps, momentum_buffer = None if self.momentum_buffer is None else self.flat.export_state(self.momentum_buffer), param_groups = [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups])

	def load_state_dict(self, sd):
		self.steps = sd['steps']
		if self.momentum_buffer is not None and sd.get('momentum_buffer') is not None:
			self.flat.import_state(self.momentum_buffer, sd['momentum_buffer'], 'SGD.load_state_dict(momentum_buffer)')
		for g, s in zip(self.param_groups, sd['param_groups']):
			g.update(s)


class _NonFinite:
	"""`not isfinite(t[i])`, evaluated (with the stream joins and the host read it takes) only when somebody asks: bool(flag)."""

	def __init__(self, t, i, engine = None):
		self.t, self.i, self.engine = t, i, engine

	def __bool__(self):
		if self.engine is not None:
			self.engine.join_comm_stream()
		return not bool(torch.isfinite(self.t[self.i]))


def train_step(model, optimizer, x, xlen, y, ylen, max_norm = 100.0, accumulate_iterations = 1, iteration = 0, world_size = 1, sync_metrics = True, device_gate = True):
	"""One iteration of the reference loop, train.py:745-783.

	Returns dict(loss, loss_cur, entropy, grad_norm, skipped) of 0-d device tensors (read them lazily: no forced host sync).
	The reference's inf/NaN gate (train.py:769: skip backward and the optimizer step) is applied on the device when
	device_gate is set and there is no gradient accumulation: backward still runs, but the fused optimizer kernel reads the
	(all-reduced) loss and leaves parameters and momentum untouched when it is not finite -- the same parameters as the
	reference after the iteration, without draining the GPU in the middle of every step.  With accumulation (or
	device_gate = False) the gate is the reference's host-side check."""
	Fn.begin_step(x.device)  # the device-resident dropout key of this step (one single-thread launch; the same whether the step is launched eagerly or replayed from a graph)
	out = model(x, xlen, y = y, ylen = ylen)
	log_probs, olen, loss_vec = out['log_probs'], out['olen'], out['loss']
	# train.py:754-756 in one launch (ops.loss_head): loss = mean(loss_vec * ylen[:, 0]) / accum, loss_cur = mean(loss_vec), the mean
	# entropy, the inf/NaN flag, and d loss / d loss_vec (the vector backward() is seeded with)
	ent = M.entropy(log_probs[0].detach(), olen[0], dim = 1)
	scaler = getattr(getattr(optimizer, 'flat', None), 'loss_scaler', None)  # fp16: backward is seeded with the scaled loss's gradient (train.py:770-772)
	engine = model if hasattr(model, 'finish_gradient_sync') else None
	group = engine.group if engine is not None else None
	# The two scalar all-reduces of train.py:759-760 as one 2-element all-reduce.  With a data-parallel engine that runs
	# collectives it is NOT optional: the skip gate below must be the same number on every rank (the gradients are already
	# averaged; a rank-local inf/NaN would make one replica skip the update the others apply).
	reduce_metrics = (engine is not None and engine.collectives) or (sync_metrics and torch.distributed.is_available() and torch.distributed.is_initialized() and (world_size > 1 or sync_metrics is True))
	ranks = torch.distributed.get_world_size(group) if reduce_metrics else 1
	scalars, grad_loss_vec, skipped = ops.loss_head(loss_vec, ylen[:, 0], ent, accumulate_iterations, loss_scaler = None if scaler is None else scaler.current, metric_scale = 1.0 / ranks)
	skipped = skipped[0]
	if reduce_metrics:
		# SUM of (local mean / world) = the mean over ranks, with no divide kernel behind it.  With an engine the collective runs on the
		# engine's communication stream like the gradient buckets (the fused optimizer step, which reads the reduced loss as its skip
		# gate, is ordered behind that stream by finish_gradient_sync); the host-visible flag is evaluated only if somebody looks.
		if engine is not None and engine.collectives:
			engine.all_reduce_async(scalars[1:3])
		else:
			torch.distributed.all_reduce(scalars[1:3], op = torch.distributed.ReduceOp.SUM, group = group)
		skipped = _NonFinite(scalars, 1, engine)
	loss_cur = scalars[1]
	res = dict(loss = scalars[0], loss_cur = loss_cur, entropy = scalars[2], grad_norm = None, skipped = skipped)
	gate = None
	if device_gate and accumulate_iterations == 1 and hasattr(optimizer, 'flat'):
		gate = scalars[1:2]
	elif bool(skipped):
		res['skipped'] = True
		return res
	last_of_group = iteration % accumulate_iterations == 0
	if engine is not None and not last_of_group:
		with engine.no_sync():
			loss_vec.backward(grad_loss_vec)
	else:
		loss_vec.backward(grad_loss_vec)
	Fn.join_side_streams()  # weight gradients computed on the side stream are final from here on
	if last_of_group:
		if engine is not None:
			engine.finish_gradient_sync()
		flat = optimizer.flat
		own_norm = getattr(optimizer, 'clips_in_step', False)  # (NovoGrad forms the gradient norm inside its fused step: no separate reduction pass over the arena)
		if own_norm:
			flat.finalize_grads()
			flat.clip = (None, float(max_norm))
		else:
			res['grad_norm'] = flat.clip_grad_norm_(max_norm)
		if gate is not None:
			optimizer.step(loss_gate = gate)
		else:
			optimizer.step()
		if own_norm:
			res['grad_norm'] = optimizer.total_norm[0]
		optimizer.zero_grad()
	return res


_HIP_NODE_TYPES = {0: 'kernel', 1: 'memcpy', 2: 'memset', 3: 'host', 4: 'graph', 5: 'empty', 6: 'wait_event', 7: 'event_record', 8: 'ext_sem_signal', 9: 'ext_sem_wait', 10: 'mem_alloc', 11: 'mem_free', 12: 'memcpy_from_symbol', 13: 'memcpy_to_symbol'}  # hipGraphNodeType
_hip_runtime = []


def capture_node_kinds(device):
	"""{node type: count} of the graph being captured on the current stream of `device` (hipStreamGetCaptureInfo_v2 -> hipGraphGetNodes ->
	hipGraphNodeGetType, over the HIP runtime torch has loaded), or None when the stream is not capturing / the runtime cannot be asked."""
	import ctypes
	try:
		if not _hip_runtime:
			_hip_runtime.append(ctypes.CDLL('libamdhip64.so'))
		hip = _hip_runtime[0]
		status, gid, graph, deps, ndeps = ctypes.c_int(0), ctypes.c_ulonglong(0), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t(0)
		stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
		if hip.hipStreamGetCaptureInfo_v2(stream, ctypes.byref(status), ctypes.byref(gid), ctypes.byref(graph), ctypes.byref(deps), ctypes.byref(ndeps)) != 0 or status.value != 1 or not graph:
			return None
		n = ctypes.c_size_t(0)
		if hip.hipGraphGetNodes(graph, None, ctypes.byref(n)) != 0:
			return None
		nodes = (ctypes.c_void_p * max(n.value, 1))()
		if hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n)) != 0:
			return None
		kinds = {}
		for i in range(n.value):
			t = ctypes.c_int(-1)
			hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
			name = _HIP_NODE_TYPES.get(t.value, str(t.value))
			kinds[name] = kinds.get(name, 0) + 1
		return kinds
	except (OSError, AttributeError):
		return None


class GraphedTrainStep:
	"""train_step(model, optimizer, x, xlen, y, ylen) with accumulate_iterations = 1, replayed from a HIP graph: one graph per batch shape
	(x.shape, y.shape -- a bucketed loader that pads every batch of a bucket to the bucket's ceiling, datasets.bucket_ceiling, produces
	about a dozen), captured after `warmup` eager steps of that shape.  A JasperNetLarge step is ~530 kernel launches and ~19 ms of
	Python for ~40 ms of GPU work; a replay is one call.

	What makes the captured step a faithful replay (tests/test_full_size_and_step_graphs_gpu.py compares 20 / 24 steps bit for bit with the eager path):
	* everything that changes from step to step is read from device memory when the kernels RUN -- the dropout step key
	  (functional.begin_step), the loss scaler, the optimizer's step counters / EMAs, the learning rate (`optimizer.lr_dev`, refreshed here
	  whenever the host's scheduler changed param_groups[0]['lr']), the device-side skip gates;
	* double-buffered device state is handed back to the buffer the graph reads (LossScaler.advance, optimizers._advance);
	* every buffer a captured kernel touches is either persistent (parameters, arenas, optimizer state, packed weights) or was allocated
	  inside the capture, from a memory pool all graphs of this object share (they never run concurrently): functional.CAPTURING turns the
	  per-module / per-stream caches off;
	* the capture is one chain of nodes (linear = True: side streams off while capturing; linear = False forks and joins the weight-gradient
	  side stream and the dgrad-weight prepack stream inside the capture -- slower on ROCm 7.2, see _capture).

	Inputs are copied into the graph's static buffers (one device-to-device copy each; pass the static buffers themselves -- .inputs(key)
	-- to skip it).  Returned metrics are the graph's static output tensors: read them before the next call with the same shape.
	A data-parallel engine is captured WITH its collectives when they are RCCL's (parallel.DataParallelEngine.capturable): the bucket
	all-reduces are recorded on the communication stream, which forks off the capturing stream at the buckets' ready events and rejoins it in
	finish_gradient_sync, so a replayed step overlaps the exchange with the backward pass like the eager one and costs the host one launch
	(eagerly every rank spends ~4 / ~19 ms of Python per Wav2Letter / JasperNetLarge step).  gloo engines (host waits) stay eager: the call
	falls through to train_step (with this object's world_size / sync_metrics)."""

	def __init__(self, model, optimizer, max_norm = 100.0, warmup = 1, enabled = True, linear = os.environ.get('CONVASR_GRAPH_FORKED') != '1', max_graphs = 64, world_size = 1, sync_metrics = True):
		self.model, self.optimizer, self.max_norm, self.warmup, self.linear = model, optimizer, max_norm, max(int(warmup), 1), linear
		self.world_size, self.sync_metrics = world_size, sync_metrics  # (for the eager fall-through of a data-parallel engine: what train_step averages the logged metrics over)
		self.max_graphs = max_graphs  # batch shapes beyond this many stay eager (a loader that does not pad to bucket ceilings produces a new shape per batch: every capture keeps its static inputs and outputs alive)
		engine = model if hasattr(model, 'finish_gradient_sync') else None
		self.enabled = bool(enabled) and (engine is None or engine.capturable)  # (an engine whose collectives are RCCL's is captured with them: the communication stream forks off and rejoins inside the graph; gloo waits on the host and stays eager)
		if self.enabled and engine is not None and engine.collectives:
			engine.enable_capture()  # (a rendezvous of all ranks: the engine's own RCCL communicator, through which the captured collectives go -- parallel.DataParallelEngine.enable_capture)
		self.graphs, self.seen = {}, {}
		self.pool = None
		self.epoch = Fn.structure_epoch()
		self.replays = self.captures = self.eager_steps = 0
		self.non_kernel_nodes = os.environ.get('CONVASR_GRAPH_FENCE') == '1'  # does any captured step hold a memset / memcpy / ... node (see _fence_transition)?
		self._lr = None

	@staticmethod
	def key_of(x, xlen, y, ylen):
		return (tuple(x.shape), x.dtype, tuple(xlen.shape), tuple(y.shape), y.dtype, tuple(ylen.shape))

	def inputs(self, key):
		g = self.graphs.get(key)
		return None if g is None else g['static']

	def _fence_transition(self, device, eager):
		"""A host wait for the stream wherever an eagerly launched step meets a replayed one (either order) -- armed only when a captured step
		contains a node that is not a kernel.  Round 6 (profiles/r06_interleave_race.txt): with the order A A B A B B C A B C A on a small Wav2Letter
		(AdamW, bf16, max_graphs = 2: C stays eager) the replay of A after the second eager C computed a different loss in most runs.  Root cause:
		the MEMSET node the capture recorded for convasr_signal_absmax's hipMemsetAsync (the first node of the step).  On ROCm 7.2 it is not
		reliably ordered against the graph's own next kernel (the atomicMax reduction into the words it clears) when the replay follows eagerly
		launched work; the utterances' maxima then come out cleared and normalize_signal scales by 1 / eps.  With that clear (and convasr_copy,
		a hipMemcpyAsync before) turned into kernels every captured step consists of kernel nodes only (tests/test_full_size_and_step_graphs_gpu.py
		counts them) and the sequence is bit-identical to the eager run with no fence (6 of 6; 2 of 6 with the memset node back in).  The fence stays
		as the guard for graphs that do contain memset / memcpy nodes (a user module's torch ops may record them): one host wait per transition,
		nothing on the replay-to-replay path."""
		if self.graphs and self.non_kernel_nodes and getattr(self, '_last_eager', None) is not None and self._last_eager != eager and device.type == 'cuda':
			torch.cuda.current_stream(device).synchronize()
		self._last_eager = eager

	def _eager(self, x, xlen, y, ylen, iteration):
		self.eager_steps += 1
		self._fence_transition(x.device, eager = True)
		return train_step(self.model, self.optimizer, x, xlen, y, ylen, max_norm = self.max_norm, iteration = iteration, world_size = self.world_size, sync_metrics = self.sync_metrics)

	def _sync_lr(self):
		lr = float(self.optimizer.param_groups[0]['lr'])
		if self.optimizer.lr_dev is None:
			self.optimizer.lr_dev = torch.empty(1, dtype = torch.float32, device = self.optimizer.flat.data.device)
			self._lr = None
		if lr != self._lr:
			self.optimizer.lr_dev.fill_(lr)  # (outside the graph, only when the scheduler moved: one tiny launch)
			self._lr = lr

	def _capture(self, key, x, xlen, y, ylen, iteration):
		dev = x.device
		static = tuple(t.clone() for t in (x, xlen, y, ylen))
		if self.pool is None:
			self.pool = torch.cuda.graph_pool_handle()
		graph = torch.cuda.CUDAGraph()
		opt = self.optimizer
		steps0 = opt.steps
		self._sync_lr()
		net = M.master_module(self.model)
		cached = getattr(net, '_dgrad_weights', None)
		if cached is not None and cached[0] == Fn.structure_epoch():
			Fn.prewarm_dgrad_pack(cached[1], net.compute_dtype)  # (a host-to-device table copy: not possible inside the capture)
		# double-buffered device state (NovoGrad's EMAs, AdamW's applied-step counter, the loss scaler): the rows stop swapping for good --
		# this graph, and every later eager step, hands the new state back to the row that is current now
		opt._pinned = True
		Fn.GRAPHS_CAPTURED[0] = True
		if getattr(opt.flat, 'loss_scaler', None) is not None:
			opt.flat.loss_scaler.pinned = True
		# every version-keyed packed copy (padded-Cout forward operands, the folded prologue, the padded head, fp32 / split operands) is
		# declared stale, so that ALL per-step pack launches are recorded in the graph: a copy that happens to be current now -- a validation
		# forward ran since the last optimizer step -- would otherwise get no pack node and go stale under replays
		Fn.force_repack()
		torch.cuda.synchronize(dev)
		# The captured step is ONE chain of nodes unless linear = False: the weight-gradient side stream and the dgrad-weight prepack stream are
		# switched off for the capture.  Measured on ROCm 7.2 (profiles/r05_graph_ab.json): a graph with forked branches replays 2 % SLOWER
		# than the same nodes in one chain (the runtime spreads the branches over several queues and pays for the cross-queue signals; with
		# DEBUG_HIP_FORCE_GRAPH_QUEUES=1 the forked capture matches the linear one), and never reaches the eager side stream's overlap.
		side, prepack = dict(Fn._side_streams), Fn.PREPACK
		if self.linear:
			Fn.join_side_streams()
			for k in Fn._side_streams:
				Fn._side_streams[k] = None
			Fn.PREPACK = False
		Fn.CAPTURING[0] = True
		try:
			with torch.cuda.graph(graph, pool = self.pool):
				res = train_step(self.model, opt, *static, max_norm = self.max_norm, iteration = iteration, world_size = self.world_size, sync_metrics = self.sync_metrics)
				kinds = capture_node_kinds(dev)
		finally:
			Fn.CAPTURING[0] = False
			Fn._side_streams.update(side)
			Fn.PREPACK = prepack
		opt.steps = steps0  # (nothing ran: the replay below is this step)
		self.graphs[key] = dict(graph = graph, static = static, res = res, node_kinds = kinds)
		if kinds is None or any(k != 'kernel' for k in kinds):
			self.non_kernel_nodes = True  # (arms _fence_transition; a capture that cannot be inspected is treated like one that holds other nodes)
		self.captures += 1

	def __call__(self, x, xlen, y, ylen, iteration = 0):
		if not self.enabled:
			return self._eager(x, xlen, y, ylen, iteration)
		if self.epoch != Fn.structure_epoch():  # conv modules were replaced (fuse_conv_bn_eval): the captured pointers are stale
			self.graphs, self.seen, self.epoch = {}, {}, Fn.structure_epoch()
		key = self.key_of(x, xlen, y, ylen)
		g = self.graphs.get(key)
		if g is None:
			n = self.seen.get(key, 0)
			if n < self.warmup or self.optimizer.steps == 0 or len(self.graphs) >= self.max_graphs:
				self.seen[key] = n + 1
				return self._eager(x, xlen, y, ylen, iteration)
			self._capture(key, x, xlen, y, ylen, iteration)
			g = self.graphs[key]
		for dst, src in zip(g['static'], (x, xlen, y, ylen)):
			if dst.data_ptr() != src.data_ptr():
				_lib.call('convasr_copy', _lib.ptr(src if src.is_contiguous() else src.contiguous()), _lib.ptr(dst), dst.numel() * dst.element_size(), _lib.stream_ptr())
		self._sync_lr()
		opt = self.optimizer
		Fn.join_prepack(x.device)  # (side streams an eager step may have left running are joined before the graph is launched)
		self._fence_transition(x.device, eager = False)
		Fn.join_side_streams()
		opt.flat.mirror_carried_over(lambda p16: g['graph'].replay())  # (the same version bookkeeping an eager optimizer launch gets: the 16-bit mirror stays current, other packed copies go stale)
		opt.steps += 1
		self.replays += 1
		return g['res']


def train_epoch(model, optimizer, batches, sampler = None, scheduler = None, iteration = 0, world_size = 1, max_norm = 100.0, accumulate_iterations = 1, max_iterations = None, on_step = None, graphs = None):
	"""The body of the reference's epoch loop (train.py:739-808) over an iterable of (meta, s, x, xlen, y, ylen) batches whose
	tensors are already on the device (convasr_amd.datasets.gpu_batches): train_step, scheduler.step(iteration) after every
	optimizer step (train.py:783), iteration += 1, sampler.batch_idx += world_size (train.py:807-808, what makes a resumed
	epoch start where it stopped).  Metrics stay on the device; on_step(iteration, batch, result) may read them.
	Returns the next iteration number."""
	for meta, s, x, xlen, y, ylen in batches:
		if graphs is not None and accumulate_iterations == 1 and world_size == 1:
			res = graphs(x, xlen, y, ylen, iteration = iteration)  # a GraphedTrainStep(model, optimizer, max_norm): replayed per batch shape
		else:
			res = train_step(model, optimizer, x, xlen, y, ylen, max_norm = max_norm, accumulate_iterations = accumulate_iterations, iteration = iteration, world_size = world_size, sync_metrics = world_size > 1)
		# train.py:769-783: the scheduler steps only inside the not-skipped branch.  Where the host knows the verdict (a host-side gate
		# returned skipped = True) the reference's behaviour is reproduced exactly; with the device-side gate the host does not wait
		# for it, and after a skipped iteration the next step's lr is one scheduler tick ahead of the reference's
		if scheduler is not None and iteration % accumulate_iterations == 0 and res['skipped'] is not True:
			scheduler.step(iteration)
		if on_step is not None:
			on_step(iteration, (meta, s, x, xlen, y, ylen), res)
		iteration += 1
		if sampler is not None:
			sampler.batch_idx += world_size
		if max_iterations is not None and iteration >= max_iterations:
			break
	return iteration
