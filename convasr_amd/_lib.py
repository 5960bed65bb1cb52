"""ctypes binding of libconvasr_hip.so (the C ABI declared in include/convasr_hip.h).

torch is imported first on purpose: it loads its own libamdhip64.so (soname libamdhip64.so.7); our library's DT_NEEDED entry
of the same soname then resolves to that already-loaded runtime, so torch's streams and device pointers are valid inside
our kernels.  There is NO fallback: if the shared library is missing the import fails loudly."""
import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CONVASR_HIP_LIB') or os.path.join(_HERE, 'libconvasr_hip.so')  # (the override is a measurement hook: an alternate build of the same sources, see build.py)

F32, BF16, I16, F16 = 0, 1, 2, 3
LOSS_SCALER_FLOATS = 8  # include/convasr_hip.h: CONVASR_LOSS_SCALER_FLOATS
ACT_NONE, ACT_RELU, ACT_HARDTANH, ACT_LEAKY_RELU = 0, 1, 2, 3
PACK_FWD, PACK_DGRAD = 0, 1
W_REFERENCE, W_KMAJOR = 0, 1  # include/convasr_hip.h: memory layout of a (Cout, Cin, K) parameter / gradient

c_int, c_i64, c_u64, c_f32, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_void_p

_SIGNATURES = dict(
	convasr_abi_version = (c_int, []),
	convasr_last_error = (ctypes.c_char_p, []),
	convasr_convert_layout = (c_int, [c_p, c_int, c_i64, c_i64, c_i64, c_p, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_p]),
	convasr_signal_absmax = (c_int, [c_p, c_int, c_int, c_int, c_p, c_p]),
	convasr_logmel_fwd = (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_f32, c_p]),
	convasr_instnorm_fwd = (c_int, [c_p, c_int, c_i64, c_i64, c_i64, c_p, c_int, c_i64, c_i64, c_i64, c_p, c_int, c_int, c_int, c_int, c_f32, c_p]),
	convasr_instnorm_running_fwd = (c_int, [c_p, c_int, c_i64, c_i64, c_i64, c_p, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_int, c_int, c_f32, c_p, c_p, c_p, c_f32, c_int, c_p, c_p]),
	convasr_output_lengths = (c_int, [c_p, c_int, c_int, c_p, c_p]),
	convasr_conv_cout_pad = (c_int, [c_int]),
	convasr_pack_conv_weight = (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_fold2_geometry = (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
	convasr_fold2_pack_weight = (c_int, [c_p, c_int, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_fold2_unfold_wgrad = (c_int, [c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_conv1d_fwd = (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_int, c_f32, c_f32, c_p, c_p, c_p]),
	convasr_conv_stats_max_rows = (c_int, [c_int, c_int]),
	convasr_grouped_conv1d_fwd = (c_int, [c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_grouped_conv1d_dgrad = (c_int, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_grouped_conv1d_wgrad_workspace_bytes = (c_i64, [c_int, c_int, c_int, c_int, c_int]),
	convasr_grouped_conv1d_wgrad = (c_int, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_reduce_rows = (c_int, [c_p, c_int, c_int, c_p, c_p]),
	convasr_debug_set_conv_v2 = (c_int, [c_int]),
	convasr_conv1d_wgrad_workspace_bytes = (c_i64, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
	convasr_conv1d_wgrad = (c_int, [c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_bn_finalize = (c_int, [c_p, c_int, c_i64, c_p, c_p, c_p, c_p, c_f32, c_f32, c_p, c_p, c_p, c_p, c_int, c_p, c_p]),
	convasr_bn_eval_scale_shift = (c_int, [c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_int, c_p]),
	convasr_bn_act_fwd = (c_int, [c_p, c_p, c_int, c_p, c_p, c_int, c_p, c_p, c_p, c_int, c_f32, c_f32, c_f32, c_u64, c_u64, c_p, c_p, c_int, c_int, c_int, c_p, c_p]),
	convasr_bn_bwd_workspace_bytes = (c_i64, [c_int, c_int, c_int]),
	convasr_bn_act_bwd_reduce = (c_int, [c_p, c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_f32, c_f32, c_f32, c_u64, c_u64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p, c_p]),
	convasr_bn_act_bwd_apply = (c_int, [c_p, c_p, c_p, c_int, c_p, c_int, c_p, c_p, c_int, c_f32, c_f32, c_f32, c_u64, c_u64, c_p, c_p, c_int, c_int, c_int, c_p, c_p]),
	convasr_bn_bwd_apply = (c_int, [c_p, c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p]),
	convasr_log_softmax_fwd = (c_int, [c_p, c_p, c_i64, c_int, c_p]),
	convasr_log_softmax_bwd = (c_int, [c_p, c_p, c_p, c_i64, c_int, c_p]),
	convasr_ctc_workspace_bytes = (c_i64, [c_int, c_int, c_int]),
	convasr_ctc_loss = (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_scale_rows = (c_int, [c_p, c_p, c_p, c_i64, c_p, c_int, c_i64, c_p]),
	convasr_loss_head = (c_int, [c_p, c_p, c_i64, c_p, c_int, c_f32, c_p, c_p, c_p, c_p, c_f32, c_p]),
	convasr_entropy = (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_f32, c_p]),
	convasr_weighted_mean_entropy = (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_f32, c_p]),
	convasr_argmax = (c_int, [c_p, c_p, c_i64, c_int, c_p]),
	convasr_sumsq_workspace_bytes = (c_i64, []),
	convasr_sumsq = (c_int, [c_p, c_i64, c_p, c_p, c_p, c_f32, c_p, c_p]),
	convasr_sgd_step = (c_int, [c_p, c_p, c_p, c_p, c_i64, c_p, c_f32, c_f32, c_f32, c_f32, c_int, c_int, c_p, c_f32, c_p, c_int, c_p, c_p, c_p, c_p]),
	convasr_adamw_step = (c_int, [c_p, c_p, c_p, c_p, c_i64, c_p, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_p, c_p, c_p, c_f32, c_p, c_int, c_p, c_p, c_p, c_p]),
	convasr_conv1d_dgrad_bn_reduce = (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_p, c_int, c_f32, c_f32, c_f32, c_u64, c_u64, c_p, c_p, c_p, c_p, c_p, c_p]),
	convasr_bn_bwd_finalize = (c_int, [c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_i64, c_int, c_p]),
	convasr_novograd_step = (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_i64, c_p, c_int, c_p, c_p, c_f32, c_f32, c_f32, c_f32, c_f32, c_f32, c_int, c_int, c_p, c_p, c_f32, c_p, c_int, c_p, c_p, c_p, c_p]),
	convasr_step_begin = (c_int, [c_p, c_p]),
	convasr_conv1x1_grouped = (c_int, [c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, ctypes.POINTER(c_int), c_p]),
	convasr_add16 = (c_int, [c_p, c_p, c_p, c_i64, c_int, c_p]),
	convasr_bn_bwd_reduce_many_workspace_bytes = (c_i64, [c_int, c_int, c_int, c_int]),
	convasr_bn_bwd_reduce_many = (c_int, [c_p, c_p, c_f32, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p]),
	convasr_pack_dgrad_item_bytes = (c_int, []),
	convasr_pack_dgrad_grouped = (c_int, [c_p, c_int, c_int, c_p]),
	convasr_bn_finalize_grouped = (c_int, [c_int, c_p, c_int, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_p]),
	convasr_bn_bwd_finalize_grouped = (c_int, [c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_int, c_p]),
	convasr_bn_bwd_apply_grouped = (c_int, [c_p, c_int, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_p]),
	convasr_wgrad1x1_grouped_workspace_bytes = (c_i64, [c_int, c_p, c_p, c_int, c_int]),
	convasr_wgrad1x1_grouped = (c_int, [c_int, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_p]),
	convasr_copy = (c_int, [c_p, c_p, c_i64, c_p]),
	convasr_colsum_workspace_bytes = (c_i64, [c_i64, c_int]),
	convasr_colsum = (c_int, [c_p, c_int, c_i64, c_int, c_p, c_p, c_int, c_p]),
	convasr_cast_scale = (c_int, [c_p, c_int, c_p, c_int, c_i64, c_f32, c_p]),
	convasr_bn_act_fwd_split3 = (c_int, [c_p, c_p, c_int, c_p, c_p, c_int, c_p, c_p, c_p, c_int, c_f32, c_f32, c_f32, c_u64, c_u64, c_p, c_p, c_int, c_int, c_int, c_p, c_p]),
	convasr_bn_act_bwd_apply_split3 = (c_int, [c_p, c_p, c_p, c_int, c_p, c_int, c_p, c_p, c_int, c_f32, c_f32, c_f32, c_u64, c_u64, c_p, c_p, c_int, c_int, c_int, c_p, c_p]),
	convasr_split3 = (c_int, [c_p, c_p, c_int, c_i64, c_int, c_int, c_p]),
	convasr_pack_conv_weight_split3 = (c_int, [c_p, c_int, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_bn_act_bwd_apply_to_half = (c_int, [c_p, c_p, c_p, c_int, c_p, c_int, c_p, c_p, c_int, c_f32, c_f32, c_f32, c_u64, c_u64, c_p, c_p, c_int, c_int, c_int, c_p, c_p]),
	convasr_conv1d_fwd_splitk_plan = (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_conv1d_fwd_splitk = (c_int, [c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_int, c_f32, c_f32, c_p, c_int, c_p, c_p]),
	convasr_conv1d_wgrad_ld_supported = (c_int, [c_int] * 10),
	convasr_conv1d_wgrad_ld = (c_int, [c_p, c_int, c_p, c_int, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_p]),
	convasr_novograd_item_elems = (c_i64, []),
	convasr_collate_pad = (c_int, [c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_i64, c_p]),
	convasr_ctc_alignment_workspace_bytes = (c_i64, [c_int, c_int, c_int]),
	convasr_ctc_alignment = (c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_int, c_int, c_int, c_int, c_int, c_p]),
)

_lib = None


class ConvasrHipError(RuntimeError):
	pass


def declared_symbols():
	return sorted(_SIGNATURES)


def load():
	"""Load the shared library (once) and type every entry point.  Raises if it is missing -- no CPU fallback exists."""
	global _lib
	if _lib is None:
		if not os.path.exists(LIB_PATH):
			raise ConvasrHipError(f'{LIB_PATH} not found: build it with `python -m convasr_amd.build` (hipcc --offload-arch=gfx950)')
		lib = ctypes.CDLL(LIB_PATH)
		if hasattr(lib, 'convasr_debug_read_stamps'):  # diagnostic builds only (build.py --variant ... -DCONVASR_STAMPS=1)
			lib.convasr_debug_read_stamps.restype, lib.convasr_debug_read_stamps.argtypes = c_int, [c_p, c_int]
			lib.convasr_debug_read_wgrad_stamps.restype, lib.convasr_debug_read_wgrad_stamps.argtypes = c_int, [c_p, c_int]
		for name, (res, args) in _SIGNATURES.items():
			fn = getattr(lib, name)
			fn.restype, fn.argtypes = res, args
		if lib.convasr_abi_version() != 10:
			raise ConvasrHipError('ABI version mismatch')
		_lib = lib
	return _lib


_fns = {}
calls = [0]  # C-ABI calls made so far (bench.py reads the difference over a step: what a step costs the host in calls, eager vs replayed)


def call(name, *args):
	fn = _fns.get(name)
	if fn is None:
		fn = _fns[name] = getattr(load(), name)
	calls[0] += 1
	rc = fn(*args)
	if rc != 0:
		raise ConvasrHipError(f'{name} failed ({rc}): {load().convasr_last_error().decode()}')


def call_rc(name, *args):
	"""For entry points whose positive return codes are answers, not errors (negative codes still raise)."""
	lib = load()
	calls[0] += 1
	rc = getattr(lib, name)(*args)
	if rc < 0:
		raise ConvasrHipError(f'{name} failed ({rc}): {lib.convasr_last_error().decode()}')
	return rc


class _TimingEvent:
	"""A HIP event made for TIMING only: hipEventDisableSystemFence -- recording it does not write back and invalidate the caches the way a
	default event's system-scope fence does ("can improve the accuracy of timing measurements by avoiding the cost of cache writeback and
	invalidation, and the performance impact of those actions on the execution of following work", hip_runtime_api.h).  The default events
	torch.cuda.Event makes cost the benchmark's timed region 1.6 % around 34 launches per step (profiles/r06_event_cost.txt)."""
	_hip = None
	FLAGS = int(os.environ.get('CONVASR_TIMER_EVENT_FLAGS', '0x20000000'), 0)  # hipEventDisableSystemFence

	@classmethod
	def runtime(cls):
		if cls._hip is None:
			try:
				hip = ctypes.CDLL('libamdhip64.so')
				hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
				hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
				hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
				hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
				cls._hip = hip
			except (OSError, AttributeError):
				cls._hip = False
		return cls._hip or None

	def __init__(self):
		self.handle = ctypes.c_void_p()
		if self.runtime().hipEventCreateWithFlags(ctypes.byref(self.handle), self.FLAGS) != 0:
			raise ConvasrHipError('hipEventCreateWithFlags failed')

	def record(self):
		if self._hip.hipEventRecord(self.handle, stream_ptr()) != 0:
			raise ConvasrHipError('hipEventRecord failed')

	def elapsed_time(self, end):
		ms = ctypes.c_float(0.0)
		if self._hip.hipEventElapsedTime(ctypes.byref(ms), self.handle, end.handle) != 0:
			raise ConvasrHipError('hipEventElapsedTime failed (were the events recorded and the stream synchronised?)')
		return ms.value

	def __del__(self):
		if self.handle and self._hip:
			self._hip.hipEventDestroy(self.handle)
			self.handle = None


class KernelTimer:
	"""HIP-event timing of selected launches on the stream they are launched on (bench.py's roofline leg).  Events are
	recorded around the C-ABI call; elapsed times are read after the caller synchronises.  The events are timing-only ones
	(_TimingEvent: no system-scope fence at the record) unless CONVASR_TIMER_EVENTS=torch or the HIP runtime cannot be reached."""

	def __init__(self, only = None):
		self.records = {}
		self.only = None if only is None else set(only)  # families to time; launches of other families run untouched (an event pair costs stream time)
		self.sequence = []  # (family, symbol class) of EVERY launch that came through, in launch order (bench.py maps a profiler's dispatch list onto it)
		self.raw = os.environ.get('CONVASR_TIMER_EVENTS', 'raw') != 'torch' and _TimingEvent.runtime() is not None
		self.event_kind = 'hipEventDisableSystemFence' if self.raw else 'torch.cuda.Event'

	def timed(self, family, work, fn, nbytes = 0.0, symbol = None):
		self.sequence.append((family, symbol))
		if self.only is not None and family not in self.only:
			return fn()
		start, end = (_TimingEvent(), _TimingEvent()) if self.raw else (torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True))
		start.record()
		fn()
		end.record()
		self.records.setdefault(family, []).append((start, end, work, nbytes))

	def summary(self):
		out = {}
		for family, recs in self.records.items():
			ms = [s.elapsed_time(e) for s, e, _, _ in recs]
			out[family] = dict(launches = len(recs), total_ms = sum(ms), avg_us = 1e3 * sum(ms) / max(len(ms), 1), work = sum(r[2] for r in recs), bytes = sum(r[3] for r in recs))
		return out


class _OrderEvents:
	"""stream_wait(dst, src): dst waits for what is enqueued on src so far -- torch's dst.wait_stream(src) with an event that carries neither a
	time stamp nor a system-scope fence (hipEventDisableTiming | hipEventDisableSystemFence): the hand-overs between the step's streams (weight
	gradients on a side stream, the dgrad-copy stream) order DEVICE work only, nothing the host inspects."""
	FLAGS = 0x2 | 0x20000000
	ring, pos = {}, {}

	@classmethod
	def wait(cls, dst, src):
		hip = _TimingEvent.runtime()
		if hip is None:
			return dst.wait_stream(src)
		dev = src.device_index
		ring = cls.ring.get(dev)
		if ring is None:
			hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
			ring = cls.ring[dev] = []
			with torch.cuda.device(dev):
				for _ in range(256):
					h = ctypes.c_void_p()
					if hip.hipEventCreateWithFlags(ctypes.byref(h), cls.FLAGS) != 0:
						raise ConvasrHipError('hipEventCreateWithFlags failed')
					ring.append(h)
			cls.pos[dev] = 0
		ev = ring[cls.pos[dev]]
		cls.pos[dev] = (cls.pos[dev] + 1) % len(ring)
		if hip.hipEventRecord(ev, ctypes.c_void_p(src.cuda_stream)) != 0 or hip.hipStreamWaitEvent(ctypes.c_void_p(dst.cuda_stream), ev, 0) != 0:
			raise ConvasrHipError('stream_wait: hipEventRecord / hipStreamWaitEvent failed')


RAW_STREAM_EVENTS = os.environ.get('CONVASR_RAW_STREAM_EVENTS') == '1'  # A/B hook (round 6): the side-stream hand-overs through fence-free events


def stream_wait(dst, src):
	if RAW_STREAM_EVENTS and not torch.cuda.is_current_stream_capturing():
		_OrderEvents.wait(dst, src)
	else:
		dst.wait_stream(src)


timer = None  # set to a KernelTimer by bench.py for the timed region


def timed(family, work, fn, nbytes = 0.0, symbol = None):
	"""symbol: which kernel symbol the C side will pick, where a profiler's per-dispatch rows must be told apart ('v2s16' = the
	LDS-DMA conv kernel with 16-bit input AND output, conv1d_igemm_v2s_kernel<H, H, *>)."""
	if timer is None:
		fn()
	else:
		timer.timed(family, work, fn, nbytes, symbol)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def stream_ptr():
	"""hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object through several
	layers of Python (10 us a call, measured: ~8 ms of host time per JasperNetLarge step at 1,200 launches); the raw accessors torch
	exposes for exactly this purpose return the same handle in ~0.3 us."""
	if _raw_stream is not None and _raw_device is not None:
		return _raw_stream(_raw_device())
	return torch.cuda.current_stream().cuda_stream


def ptr(t):
	return None if t is None else t.data_ptr()


def dtype_code(dtype):
	return {torch.float32: F32, torch.bfloat16: BF16, torch.int16: I16, torch.float16: F16}[dtype]


def require_cuda(*tensors):
	for t in tensors:
		if t is not None and not t.is_cuda:
			raise ConvasrHipError('convasr_amd runs on an MI355X only: tensor on ' + str(t.device) + ' (there is no CPU path; the CPU restatement lives in oracle/ and is test infrastructure)')
