"""Node types of a captured training step (torch.cuda.CUDAGraph.debug_dump -> hipGraphDebugDotPrint): are there memset / memcpy nodes left?"""
import os, sys, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import convasr_amd as ca
import test_split_operand_gpu as T

d = torch.device('cuda:0')
which = sys.argv[1] if len(sys.argv) > 1 else 'wav2letter_bf16'
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
found = {}
NODE_TYPES = {0: 'kernel', 1: 'memcpy', 2: 'memset', 3: 'host', 4: 'graph', 5: 'empty', 6: 'wait_event', 7: 'event_record', 8: 'ext_sem_signal', 9: 'ext_sem_wait', 10: 'mem_alloc', 11: 'mem_free', 12: 'memcpy_from_symbol', 13: 'memcpy_to_symbol'}  # hipGraphNodeType
_exit = torch.cuda.graph.__exit__
def exit_hook(self, *a):
	stream = torch.cuda.current_stream().cuda_stream
	status, gid, graph, deps, ndeps = ctypes.c_int(0), ctypes.c_ulonglong(0), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t(0)
	rc = hip.hipStreamGetCaptureInfo_v2(ctypes.c_void_p(stream), ctypes.byref(status), ctypes.byref(gid), ctypes.byref(graph), ctypes.byref(deps), ctypes.byref(ndeps))
	n = ctypes.c_size_t(0)
	rc2 = hip.hipGraphGetNodes(graph, None, ctypes.byref(n))
	nodes = (ctypes.c_void_p * n.value)()
	hip.hipGraphGetNodes(graph, nodes, ctypes.byref(n))
	kinds = collections.Counter()
	for i in range(n.value):
		t = ctypes.c_int(-1)
		hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t))
		kinds[NODE_TYPES.get(t.value, t.value)] += 1
	found.update(rc = (rc, rc2), status = status.value, nodes = n.value, kinds = dict(kinds))
	return _exit(self, *a)
torch.cuda.graph.__exit__ = exit_hook
ca.functional.manual_seed(23)
torch.manual_seed(4)
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
if which == 'wav2letter_bf16':
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.1, base_width = 64, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = ca.optimizers.AdamW(flat, lr = 1e-3, weight_decay = 1e-2)
elif which == 'dense_f16':
	model = ca.models.JasperNet(64, [38], frontend = fe, base_width = 64, kernel_sizes = [11, 13, 17], out_width_factors = [2, 3, 4], dropouts = [0.2] * 3, out_width_factors_large = [4, 4], residual = 'dense', repeat = 2, num_subblocks = 2, dropout = 0.2, check_time_dim_padded = False, temporal_mask = False).to(d).train()
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
	ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2')
else:
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.1, base_width = 64, check_time_dim_padded = False, compute_dtype = which).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
stepper = ca.train.GraphedTrainStep(model, opt, warmup = 1)
batch = T._batch(d, 4, 4)
for it in range(3):
	stepper(*batch, iteration = it)
torch.cuda.synchronize()
assert stepper.captures == 1
print(which, found)
