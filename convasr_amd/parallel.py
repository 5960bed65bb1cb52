"""Data-parallel training over the GPUs of one node (reference: DistributedDataParallel at models.py:763, init at train.py:852-874).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm).  The gradient arena of FlatParameters is
cut into contiguous buckets; backward kernels report each parameter the moment its gradient is final, and as soon as every
parameter of a bucket has reported, that bucket's sum-all-reduce is enqueued asynchronously (RCCL runs it on its own HIP
stream, ordered after the producing kernels by an event) -- so communication overlaps the rest of the backward conv stack.
xGMI is point-to-point, so a few large buckets (default 64 MiB, CONVASR_BUCKET_MIB) beat many small ones; only the buckets that
complete last (the first layers, nothing left to overlap them with) are kept small: 4, 8, 16, 32 MiB.  Batch-norm statistics stay
per GPU, exactly as in the reference's training path (train.py:704 does not pass synchronize_bn).

CONVASR_COMM_THREAD=1 moves the host side of every collective (20-50 us each inside torch.distributed / RCCL) to a helper thread.
It is OFF by default: it has only ever run with one RCCL rank and with two gloo ranks sharing a GPU (+0.3 % there), and a first
multi-GPU run should not depend on a second thread issuing collectives."""
import os
import queue
import threading
import weakref

import torch
import torch.distributed as dist
import torch.nn as nn

from .train import FlatParameters


XGMI_LINK_GBS_PER_DIRECTION = 76.8  # MI355X: 7 xGMI links per GPU, ~153.6 GB/s each bidirectional (one link to every peer of an 8-GPU node)


def predict_exposed_comm(params_numel, offsets, buckets, world, step_ms, link_gbs = XGMI_LINK_GBS_PER_DIRECTION, efficiency = 0.7, latency_us = 40.0, forward_share = 0.33, tail_share = 0.04, bytes_per_element = 4):
	"""A prediction the first multi-GPU run can be wrong about (no curve has been measured: one GPU per box on this pool).

	Model.  The backward pass produces gradients from the END of the arena to its start; its time is apportioned to the parameters by
	their element counts (a conv's dgrad + wgrad FLOPs are proportional to its weight count at a fixed frame count; batch-norm vectors cost
	nothing): bucket b (arena range [lo, hi)) is complete at  t_ready(b) = step * (forward_share + backward_share * (elements at offsets >=
	lo) / all elements),  backward_share = 1 - forward_share - tail_share (tail: clip + optimizer, after the exchange).  The all-reduce of S
	bytes over the node's fully connected xGMI (a direct link to every peer, `link_gbs` GB/s per direction): reduce-scatter + all-gather
	each move S / N per link -> t = 2 (S / N) / (efficiency * link_gbs) + latency.  Collectives run one after the other on the communication
	stream in the order the buckets complete.  exposed = how long the last collective runs past the end of the backward pass (what
	finish_gradient_sync waits for).  Returns dict(exposed_comm_ms, comm_ms_total, backward_end_ms, per_bucket = [...])."""
	total = float(sum(params_numel))
	bwd_share = 1.0 - forward_share - tail_share
	t_fwd, t_bwd_end = step_ms * forward_share, step_ms * (forward_share + bwd_share)
	rows, t_free = [], 0.0
	for b in sorted(buckets, key = lambda b: -b['lo']):  # completion order: from the end of the arena
		above = sum(n for n, o in zip(params_numel, offsets) if o >= b['lo'])
		ready = t_fwd + (t_bwd_end - t_fwd) * above / total
		nbytes = (b['hi'] - b['lo']) * bytes_per_element  # (2 with a 16-bit exchange: DataParallelEngine(grad_comm_dtype))
		t = 0.0 if world <= 1 else 2.0 * (nbytes / world) / (efficiency * link_gbs * 1e9) * 1e3 + latency_us * 1e-3
		start = max(ready, t_free)
		t_free = start + t
		rows.append(dict(mib = round(nbytes / 2 ** 20, 1), ready_ms = round(ready, 3), start_ms = round(start, 3), comm_ms = round(t, 3), end_ms = round(t_free, 3)))
	return dict(exposed_comm_ms = round(max(0.0, t_free - t_bwd_end), 3), comm_ms_total = round(sum(r['comm_ms'] for r in rows), 3), backward_end_ms = round(t_bwd_end, 3), step_ms = round(step_ms, 3),
		world = world, link_gbs_per_direction = link_gbs, efficiency = efficiency, latency_us = latency_us, bytes_per_element = bytes_per_element, per_bucket = rows,
		predicted_scaling = None if world <= 1 else round(world * step_ms / (step_ms + max(0.0, t_free - t_bwd_end)), 3),
		model = 'direct reduce-scatter + all-gather over one xGMI link per peer; bucket ready times from the backward pass apportioned by parameter counts; serial collectives on the communication stream (convasr_amd.parallel.predict_exposed_comm)')


class DataParallelEngine(nn.Module):
	def __init__(self, module, device = None, bucket_bytes = int(os.environ.get('CONVASR_BUCKET_MIB', 64)) << 20, process_group = None, flat = None, force_collectives = False, first_bucket_bytes = 4 << 20, fold_mean = True, comm_thread = os.environ.get('CONVASR_COMM_THREAD', '0') == '1', measure_exposed_comm = False, grad_comm_dtype = os.environ.get('CONVASR_GRAD_COMM', 'auto')):
		"""grad_comm_dtype: what the gradient buckets travel as.  None / 'f32': the fp32 arena itself (in place).  torch.float16 / torch.bfloat16
		(or 'f16' / 'bf16'): every bucket is packed into a persistent 16-bit send buffer (x 1 / world size: the mean is formed before the sum),
		all-reduced at half the bytes and unpacked into the arena, all on the communication stream -- what the reference does under apex O2,
		whose model gradients ARE fp16 (models.py:744-762, train.py:771).  'auto' (default): fp16 when the module computes in fp16 (opt_level
		O1-O3), fp32 otherwise (bf16's 8 significant bits are not spent on gradients unless asked for)."""
		super().__init__()
		self.module = module
		self.group = process_group
		self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
		self.collectives = self.world_size > 1 or (force_collectives and dist.is_initialized())  # world 1 + force: RCCL smoke test
		self.flat = flat if flat is not None else getattr(module, '_convasr_flat', None) or FlatParameters(module)
		module._convasr_flat = self.flat
		self.buckets = self._make_buckets(bucket_bytes, min(first_bucket_bytes, bucket_bytes))
		self._pending = []
		self._ready = []  # complete buckets whose collective has not been enqueued yet: (bucket index, events recorded on the producer streams)
		# enqueueing a collective costs tens of microseconds of host time inside torch.distributed / RCCL (mostly outside the GIL): a helper
		# thread does it while this one keeps launching the backward pass's kernels
		self._jobs = queue.Queue() if (comm_thread and self.collectives and self.flat.data.is_cuda) else None
		self._worker = None
		self._comm_dirty = False  # something was enqueued on the communication stream that the main stream has not joined yet
		self._remaining = [len(b['params']) for b in self.buckets]
		self._main_stream = None  # the stream forward() ran on: dgamma / dbeta and (without a side stream) every weight gradient are produced there
		self._comm_stream = None  # collectives are issued from here, ordered after BOTH the main stream and the side (wgrad) stream
		self.fold_mean = fold_mean  # True: the 1 / world_size of the gradient mean rides in the optimizer kernel (flat.grad_scale) instead of a pass over the arena
		self.sync = True  # False inside no_sync(): gradients accumulate locally, nothing is launched (gradient accumulation)
		self.grad_comm_dtype = {None: None, 'f32': None, 'auto': 'auto', 'f16': torch.float16, 'bf16': torch.bfloat16}.get(grad_comm_dtype, grad_comm_dtype)
		self._comm_bufs = {}  # bucket index -> persistent 16-bit send / receive buffer (stable addresses: a captured step graph bakes them)
		self._mean_in_comm = False  # the last exchange already divided by the world size (16-bit buckets): finish_gradient_sync must not do it again
		self._capture_comm = None  # rccl.Communicator of this engine's own, used for the collectives of CAPTURED steps only (enable_capture)
		for bi, b in enumerate(self.buckets):
			for p in b['params']:
				p._convasr_ready = self._make_hook(bi)
		# measurement hook (bench.py): per step a HIP event pair on the main stream around the wait for the communication stream in
		# finish_gradient_sync -- the part of the gradient exchange the backward pass did not cover
		self.measure_exposed_comm = measure_exposed_comm
		self.exposed_comm_events = []
		from . import functional as Fn
		# the backward pass calls poll() right after it has enqueued a long kernel.  The registry holds the engine weakly: an engine
		# that is dropped un-registers itself (close() does it eagerly and stops the helper thread)
		ref, key = weakref.ref(self), id(self)
		def hook():
			eng = ref()
			if eng is None:
				Fn.after_long_launch_hooks.pop(key, None)
			else:
				eng.poll()
		Fn.after_long_launch_hooks[key] = hook
		self._hook_key = key
		if self.collectives:
			dist.broadcast(self.flat.data, src = 0, group = self.group)  # identical initial replicas
			for buf in module.buffers():
				dist.broadcast(buf, src = 0, group = self.group)
			from . import functional as Fn
			Fn.bump_param_epoch()  # the arena changed behind torch's version counters: packed compute copies are stale

	def close(self):
		"""Un-register from the backward pass's hook list and stop the helper thread (idempotent; also runs when the engine is collected)."""
		from . import functional as Fn
		Fn.after_long_launch_hooks.pop(getattr(self, '_hook_key', None), None)
		worker, self._worker = getattr(self, '_worker', None), None
		if worker is not None and self._jobs is not None:
			self._jobs.put(None)  # the sentinel: run() returns
			worker.join(timeout = 5)

	def destroy_capture_comm(self):
		"""Free the engine's own RCCL communicator (after the step graphs that recorded its kernels are gone; not done implicitly)."""
		comm, self._capture_comm = self._capture_comm, None
		if comm is not None:
			comm.destroy()

	def __del__(self):
		try:
			self.close()
		except Exception:
			pass

	def predict(self, world, step_ms, **kw):
		"""predict_exposed_comm for this engine's buckets (bench.py puts it next to the measured dist.exposed_comm_ms)."""
		kw.setdefault('bytes_per_element', 2 if self.comm_dtype() is not None else 4)
		return predict_exposed_comm([p.numel() for p in self.flat.params], self.flat.offsets, self.buckets, world, step_ms, **kw)

	def exposed_comm_ms(self):
		"""Mean over the recorded steps of the time the main stream spent waiting for the communication stream (call after a device
		synchronisation); None when nothing was recorded."""
		ms = [a.elapsed_time(b) for a, b in self.exposed_comm_events]
		return sum(ms) / len(ms) if ms else None

	def _make_buckets(self, bucket_bytes, first_bucket_bytes):
		"""Contiguous arena ranges.  Backward completes them from the END of the arena towards its start, so the buckets at the start
		(the first layers) are the ones whose all-reduce little or nothing overlaps: sizes are graded -- first_bucket_bytes, then doubling
		per bucket up to bucket_bytes (4, 8, 16, 32, 64, 64, ... MiB by default) -- so that the exposed tail of the exchange is a few MiB
		while the bulk of the arena, which completes early in the backward pass, goes in a few large collectives (xGMI is point-to-point:
		large messages per link)."""
		flat = self.flat
		buckets, cur = [], None
		for p, off in zip(flat.params, flat.offsets):
			end = off + p.numel()
			limit = min(bucket_bytes, first_bucket_bytes << max(len(buckets) - 1, 0))  # the limit of the bucket being filled
			if cur is None or (end - cur['lo']) * 4 > limit and cur['params']:
				cur = dict(lo = off, hi = end, params = [])
				buckets.append(cur)
			cur['params'].append(p)
			cur['hi'] = end
		return buckets

	def _make_hook(self, bi):
		def ready(param):
			if not self.sync:
				return
			self._remaining[bi] -= 1
			if self._remaining[bi] == 0:
				self._mark_ready(bi)
		return ready

	def _comm(self, dev):
		if self._comm_stream is None:
			self._comm_stream = torch.cuda.Stream(device = dev)
		return self._comm_stream

	def _mark_ready(self, bi):
		"""Every parameter of bucket bi has its gradient enqueued.  A bucket mixes gradients produced on the main stream (dgamma / dbeta,
		dgrad-side kernels) with weight gradients that functional._run_wgrad may have produced on the side stream, and this hook fires
		on whichever of the two delivered the bucket's last parameter: events are recorded on both producers NOW (cheap), the
		collective itself -- tens of microseconds of host time in torch.distributed / RCCL, during which a short-kernel stretch of the
		backward pass would drain the GPU's queue: profiles/r03_rccl_world1_trace.json -- is enqueued by poll() right after the backward
		pass has queued its next long kernel."""
		if not self.collectives:
			return
		view = self.flat.grad[self.buckets[bi]['lo']:self.buckets[bi]['hi']]
		if not view.is_cuda:
			self._launch(bi, ())
			return
		from . import functional as Fn
		dev = view.device
		producers = {s.cuda_stream: s for s in (self._main_stream, torch.cuda.current_stream(dev), Fn.side_stream(dev)) if s is not None}
		self._ready.append((bi, [s.record_event() for s in producers.values()]))

	def comm_dtype(self):
		"""The resolved exchange type: None (the fp32 arena in place) or a 16-bit torch dtype."""
		dt = self.grad_comm_dtype
		if dt == 'auto':
			from . import models as M
			dt = torch.float16 if getattr(M.master_module(self.module), 'compute_dtype', None) == torch.float16 else None
		return dt if (dt is not None and self.flat.grad.is_cuda) else None

	def exchange_bytes(self):
		"""Bytes one step's gradient exchange moves through the collectives (per rank, before the algorithm's own factor)."""
		per = 2 if self.comm_dtype() is not None else 4
		return sum((b['hi'] - b['lo']) * per for b in self.buckets)

	@property
	def capturable(self):
		"""Can a step of this engine be captured into a HIP graph?  Only when its collectives are RCCL's (stream-ordered, no host wait) and
		librccl can be called directly (rccl.py: torch.distributed's wrapper does not survive a capture on ROCm 7.2)."""
		if not self.collectives:
			return True
		from . import rccl
		return self.flat.data.is_cuda and dist.get_backend(self.group) == 'nccl' and os.environ.get('CONVASR_GRAPH_DP', '1') != '0' and rccl.library() is not None

	def enable_capture(self):
		"""Create this engine's own RCCL communicator for captured steps (a rendezvous: every rank calls it -- train.GraphedTrainStep does, from
		its constructor).  Eager steps keep torch.distributed."""
		if self.collectives and self._capture_comm is None:
			from . import rccl
			self._capture_comm = rccl.Communicator(self.flat.data.device, self.group)

	def _captured_all_reduce(self, t):
		"""SUM all-reduce on the CURRENT stream through the engine's own communicator (a plain kernel launch: what a capture records)."""
		from . import _lib
		if self._capture_comm is None:
			raise _lib.ConvasrHipError('DataParallelEngine: a step with collectives is being captured but enable_capture() was not called (train.GraphedTrainStep does it on every rank)')
		self._capture_comm.all_reduce(t, _lib.stream_ptr())

	def poll(self):
		"""Enqueue the collectives of the buckets that became complete since the last call (no-op when there are none)."""
		from . import functional as Fn
		ready, self._ready = self._ready, []
		for bi, events in ready:
			if self._jobs is not None and not Fn.capturing():  # (a capture records what THIS thread enqueues)
				self._submit(lambda bi = bi, events = events: self._launch(bi, events))
			else:
				self._launch(bi, events)

	def _submit(self, job):
		if self._worker is None:
			jobs, device, errors = self._jobs, self.flat.data.device, []  # (no reference to the engine: the thread must not keep it alive)
			def run():
				torch.cuda.set_device(device)
				while True:
					job = jobs.get()
					try:
						if job is None:
							return
						job()
					except BaseException as e:  # surfaced by _drain() on the training thread
						errors.append(e)
					finally:
						jobs.task_done()
			self._worker_errors = errors
			self._worker = threading.Thread(target = run, name = 'convasr-comm', daemon = True)
			self._worker.start()
		self._jobs.put(job)

	def _drain(self):
		"""Wait until the helper thread has enqueued everything handed to it (host-side only: nothing here waits for the GPU)."""
		if self._jobs is not None and self._worker is not None:
			self._jobs.join()
			if self._worker_errors:
				e = self._worker_errors.pop(0)
				del self._worker_errors[:]
				raise e

	def _launch(self, bi, events):
		"""All-reduce of one complete bucket.  torch.distributed orders a collective after the CURRENT stream only, so it is issued from a
		dedicated stream that first waits for the events recorded on the gradient's producer streams."""
		from . import functional as Fn
		b = self.buckets[bi]
		view = self.flat.grad[b['lo']:b['hi']]
		half = self.comm_dtype()
		if view.is_cuda and half is not None:
			# 16-bit exchange: pack (x 1 / world) -> all-reduce -> unpack, one after the other on the communication stream.  The range is widened to
			# whole 8-element groups: the arena pads every parameter to 64 elements, and the padding holds zeros
			from . import _lib
			n = -(-(b['hi'] - b['lo']) // 8) * 8
			wide = self.flat.grad[b['lo']:b['lo'] + n]
			buf = self._comm_bufs.get(bi)
			if buf is None or buf.dtype != half:
				buf = self._comm_bufs[bi] = torch.empty(n, dtype = half, device = view.device)
			comm = self._comm(view.device)
			for ev in events:
				comm.wait_event(ev)
			with torch.cuda.stream(comm):
				s = _lib.stream_ptr()
				_lib.call('convasr_cast_scale', _lib.ptr(wide), _lib.F32, _lib.ptr(buf), _lib.dtype_code(half), n, 1.0 / self.world_size, s)
				if self._capture_comm is not None:
					self._captured_all_reduce(buf)  # (an engine whose steps are captured: RCCL directly, on THIS stream, eagerly too -- see below)
				else:
					work = dist.all_reduce(buf, op = dist.ReduceOp.SUM, group = self.group, async_op = True)
					work.wait()  # RCCL: orders the communication stream behind the collective; gloo: a host wait
				_lib.call('convasr_cast_scale', _lib.ptr(buf), _lib.dtype_code(half), _lib.ptr(wide), _lib.F32, n, 1.0, s)
			self._comm_dirty = True
			self._mean_in_comm = True
			return
		if view.is_cuda:
			comm = self._comm(view.device)
			for ev in events:
				comm.wait_event(ev)
			with torch.cuda.stream(comm):
				if self._capture_comm is not None:
					# An engine whose steps are captured into HIP graphs (enable_capture) sends EVERY collective to RCCL directly (rccl.py), on the
					# communication stream -- which, under a capture, forked off the capturing stream at the bucket's ready events and rejoins it in
					# join_comm_stream.  torch.distributed's wrapper does not survive a capture here: its asynchronous form (internal stream,
					# work.wait()) makes hipStreamEndCapture segfault; its blocking form captures and replays, but the process group's watchdog thread
					# then now and then finds an event "last recorded in a capturing stream" among the Works it polls and terminates the process --
					# also when only the EAGER steps between captures go through torch (the warm-up of new batch shapes; profiles/r06_rccl_capture_probe.json,
					# profiles/r06_notes: ROCm 7.2 / torch 2.10).  Eager-only engines keep torch.distributed and its timeouts.
					self._captured_all_reduce(view)
					self._comm_dirty = True
					return
				work = dist.all_reduce(view, op = dist.ReduceOp.SUM, group = self.group, async_op = True)
			self._comm_dirty = True
		else:
			work = dist.all_reduce(view, op = dist.ReduceOp.SUM, group = self.group, async_op = True)
		self._pending.append((work, view))

	def all_reduce_async(self, t):
		"""SUM all-reduce of a small tensor produced on the current stream (the step's two logged means), on the communication stream:
		the main stream does not wait for it until finish_gradient_sync / join_comm_stream."""
		if not t.is_cuda:
			dist.all_reduce(t, op = dist.ReduceOp.SUM, group = self.group)
			return
		comm = self._comm(t.device)
		ev = torch.cuda.current_stream(t.device).record_event()

		from . import functional as Fn

		def job():
			comm.wait_event(ev)
			with torch.cuda.stream(comm):
				if self._capture_comm is not None:
					self._captured_all_reduce(t)
				else:
					dist.all_reduce(t, op = dist.ReduceOp.SUM, group = self.group)
			t.record_stream(comm)
			self._comm_dirty = True
		if self._jobs is not None and not Fn.capturing():
			self._submit(job)
		else:
			job()

	def join_comm_stream(self):
		self._drain()
		if self._comm_stream is not None and self._comm_dirty:
			torch.cuda.current_stream(self._comm_stream.device).wait_stream(self._comm_stream)
			self._comm_dirty = False

	def no_sync(self):
		"""Context manager for all but the last backward of a gradient-accumulation group (DDP.no_sync semantics)."""
		import contextlib

		@contextlib.contextmanager
		def ctx():
			prev, self.sync = self.sync, False
			try:
				yield
			finally:
				self.sync = prev
		return ctx()

	def finish_gradient_sync(self):
		"""Call after backward, before clip / optimizer: flushes buckets whose parameters got no gradient, waits for the
		collectives (stream-side) and turns sums into means (by default lazily: flat.grad_scale = 1 / world_size, which
		clip_grad_norm_ and the fused optimizer kernels apply)."""
		for bi, left in enumerate(self._remaining):
			if left > 0:
				for p in self.buckets[bi]['params']:
					if p._convasr_fresh:
						p._convasr_grad.zero_()
						p._convasr_fresh = False
				self._mark_ready(bi)
		from . import functional as Fn
		self.poll()
		self._drain()
		timed = self.measure_exposed_comm and self.flat.data.is_cuda and not Fn.capturing()  # (a timing event cannot be read back from inside a graph)
		if timed:
			ev0 = torch.cuda.Event(enable_timing = True)
			ev0.record()
		for work, view in self._pending:
			work.wait()  # RCCL: orders the CURRENT stream behind the collective (it ran on the backend's own stream), no host block; gloo: host wait
		self.join_comm_stream()
		if timed:
			ev1 = torch.cuda.Event(enable_timing = True)
			ev1.record()
			self.exposed_comm_events.append((ev0, ev1))
		if self._mean_in_comm:
			self._mean_in_comm = False  # the 16-bit buckets carried the mean already
		elif self.world_size > 1:
			if self.fold_mean:
				self.flat.grad_scale = 1.0 / self.world_size  # flat.grad holds the SUM over ranks until the optimizer consumes it
			else:
				self.flat.grad.mul_(1.0 / self.world_size)
		self._pending = []
		self._remaining = [len(b['params']) for b in self.buckets]

	def forward(self, *args, **kwargs):
		if self.flat.data.is_cuda:
			self._main_stream = torch.cuda.current_stream(self.flat.data.device)
		return self.module(*args, **kwargs)

	def state_dict(self, *args, **kwargs):
		return self.module.state_dict(*args, **kwargs)

	def load_state_dict(self, *args, **kwargs):
		return self.module.load_state_dict(*args, **kwargs)
