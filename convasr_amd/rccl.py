"""A direct binding of RCCL's all-reduce (ctypes over the librccl.so torch ships) for the ONE place torch.distributed's wrapper cannot go:
inside a HIP-graph capture.

Why.  torch's ProcessGroupNCCL wraps every collective in a Work object whose completion events its watchdog thread polls.  Under
hipStreamBeginCapture two things go wrong on ROCm 7.2 / torch 2.10 (profiles/r06_rccl_capture_probe.json, scratch/r6_rccl_capture_probe.py):
the asynchronous form (torch's internal stream, joined through work.wait()) makes hipStreamEndCapture segfault, and the blocking form --
which captures and replays correctly -- now and then leaves a Work with the watchdog, whose hipEventQuery on an event "last recorded in a
capturing stream" (hipErrorCapturedEvent) then terminates the process (1 run in 3 of tests/_dp_rccl_world1.py).  RCCL itself has no such
problem: ncclAllReduce on a stream that is being captured records its kernel like any other launch.  So the captured data-parallel step
(train.GraphedTrainStep over parallel.DataParallelEngine) issues its bucket all-reduces through a communicator of its own, created once per
engine from an ncclUniqueId that travels over the existing torch.distributed group; eager steps keep torch.distributed (its timeouts and
error handling are what a first multi-GPU run wants).  Reference: the all-reduce DistributedDataParallel performs (models.py:763, train.py:852-874).
"""
import ctypes
import os

import torch

_NCCL_DTYPES = {torch.float32: 7, torch.float16: 6, torch.bfloat16: 9, torch.float64: 8}  # rccl.h: ncclDataType_t
_NCCL_SUM = 0


class _UniqueId(ctypes.Structure):
	_fields_ = [('internal', ctypes.c_ubyte * 128)]  # NCCL_UNIQUE_ID_BYTES (raw bytes: a c_char array field would read as a NUL-terminated copy)


_lib = None


def library():
	"""librccl.so of the running torch build (already in the process when the nccl backend is up), typed once; None when it cannot be loaded."""
	global _lib
	if _lib is None:
		path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
		try:
			lib = ctypes.CDLL(path if os.path.exists(path) else 'librccl.so')
		except OSError:
			_lib = False
			return None
		lib.ncclGetUniqueId.restype, lib.ncclGetUniqueId.argtypes = ctypes.c_int, [ctypes.POINTER(_UniqueId)]
		lib.ncclCommInitRank.restype, lib.ncclCommInitRank.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
		lib.ncclAllReduce.restype, lib.ncclAllReduce.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
		lib.ncclCommDestroy.restype, lib.ncclCommDestroy.argtypes = ctypes.c_int, [ctypes.c_void_p]
		lib.ncclGetErrorString.restype, lib.ncclGetErrorString.argtypes = ctypes.c_char_p, [ctypes.c_int]
		_lib = lib
	return _lib or None


class RcclError(RuntimeError):
	pass


def _check(lib, rc, what):
	if rc != 0:
		raise RcclError(f'{what} failed ({rc}): {lib.ncclGetErrorString(rc).decode()}')


class Communicator:
	"""One RCCL communicator over the ranks of a torch.distributed group (every rank of the group must construct it: ncclCommInitRank is a
	rendezvous).  The unique id is made by the group's rank 0 and broadcast through the group itself."""

	def __init__(self, device, group = None):
		import torch.distributed as dist
		lib = library()
		if lib is None:
			raise RcclError('librccl.so could not be loaded')
		self.lib, self.device = lib, torch.device(device)
		self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
		uid = _UniqueId()
		if self.rank == 0:
			_check(lib, lib.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
		payload = [ctypes.string_at(ctypes.addressof(uid), 128) if self.rank == 0 else None]
		dist.broadcast_object_list(payload, src = dist.get_global_rank(group, 0) if group is not None else 0, group = group, device = self.device)
		assert isinstance(payload[0], bytes) and len(payload[0]) == 128
		ctypes.memmove(ctypes.addressof(uid), payload[0], 128)
		self.comm = ctypes.c_void_p()
		with torch.cuda.device(self.device):
			_check(lib, lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), 'ncclCommInitRank')
			# the first collective of a communicator sets up its channels (allocations, streams): that must not fall into a capture
			warm = torch.zeros(8, dtype = torch.float32, device = self.device)
			self.all_reduce(warm, torch.cuda.current_stream(self.device).cuda_stream)
			torch.cuda.current_stream(self.device).synchronize()

	def all_reduce(self, t, stream):
		"""In-place SUM all-reduce of a contiguous device tensor, enqueued on the raw hipStream_t `stream` (an int): a kernel launch like any
		other -- recorded when the stream is being captured."""
		assert t.is_cuda and t.is_contiguous() and t.dtype in _NCCL_DTYPES
		_check(self.lib, self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _NCCL_DTYPES[t.dtype], _NCCL_SUM, self.comm, stream), 'ncclAllReduce')

	def destroy(self):
		comm, self.comm = self.comm, None
		if comm:
			self.lib.ncclCommDestroy(comm)
