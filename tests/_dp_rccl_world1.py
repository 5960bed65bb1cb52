"""Helper of tests/test_parallel_gpu.py (run as a child process: it initialises a one-rank RCCL process group): the data-parallel training
step with REAL RCCL collectives, eagerly and replayed from step graphs (train.GraphedTrainStep captures the bucket all-reduces on the
communication stream), with the fp32 and the 16-bit gradient exchange.  Prints one JSON line."""
import json
import os
import sys

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', sys.argv[1] if len(sys.argv) > 1 else '29517')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
import torch.distributed as dist

d = torch.device('cuda:0')
torch.cuda.set_device(d)
dist.init_process_group('nccl', rank = 0, world_size = 1, device_id = d)
import convasr_amd as ca


def batches(n):
	g = torch.Generator().manual_seed(5)
	out = []
	for i in range(n):
		B, secs = [(4, 4), (3, 5)][i % 2]
		x = torch.rand(B, 16000 * secs, generator = g) * 2 - 1
		y = torch.randint(0, 37, (B, 1, 64), generator = g)
		ylen = torch.randint(10, 5 * secs, (B, 1), generator = g)
		out.append(tuple(t.to(d) for t in (x, torch.linspace(0.6, 1, B), y, ylen)))
	return out


def run(graphed, comm, dt = torch.bfloat16, opt_level = None, n = 10):
	ca.functional.manual_seed(17)
	torch.manual_seed(3)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.2, base_width = 64, check_time_dim_padded = False, compute_dtype = dt).to(d).train()
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	if opt_level is not None:
		ca.models.data_parallel_and_autocast(model, opt, opt_level = opt_level)
	engine = ca.parallel.DataParallelEngine(model, device = d, force_collectives = True, bucket_bytes = 1 << 20, first_bucket_bytes = 64 << 10, grad_comm_dtype = comm)
	assert engine.collectives and engine.capturable and len(engine.buckets) >= 4
	stepper = ca.train.GraphedTrainStep(engine, opt, warmup = 1, enabled = graphed, world_size = 1, sync_metrics = True)
	assert stepper.enabled == graphed
	trace, p0, after1 = [], flat.data.clone(), None
	for it, b in enumerate(batches(n)):
		r = stepper(*b, iteration = it)
		trace.append((float(r['loss']), float(r['loss_cur']), repr(float(r['grad_norm'])), bool(r['skipped'])))  # (repr: a NaN norm of an overflowed start-up step must compare equal to itself)
		if after1 is None and bool(torch.isfinite(r['grad_norm'])):
			after1 = (it, flat.data.clone())  # the parameters right after the first update that was applied
	torch.cuda.synchronize()
	out = dict(trace = trace, params = flat.data.clone(), captures = stepper.captures, replays = stepper.replays, node_kinds = [g['node_kinds'] for g in stepper.graphs.values()], fence_armed = stepper.non_kernel_nodes, comm_dtype = str(engine.comm_dtype()), exchange = engine.exchange_bytes(), after1 = after1, p0 = p0)
	engine.close()
	return out


res = {}
try:
	e32, g32 = run(False, None), run(True, None)
	res['fp32_exchange'] = dict(captures = g32['captures'], replays = g32['replays'], node_kinds = g32['node_kinds'], fence_armed = g32['fence_armed'], trace_equal = e32['trace'] == g32['trace'], params_equal = bool(torch.equal(e32['params'], g32['params'])), exchange_bytes = e32['exchange'])
	e16, g16 = run(False, 'auto', torch.float16, 'O2'), run(True, 'auto', torch.float16, 'O2')
	ref16 = run(False, None, torch.float16, 'O2')
	res['fp16_exchange'] = dict(comm_dtype = e16['comm_dtype'], captures = g16['captures'], replays = g16['replays'], node_kinds = g16['node_kinds'], fence_armed = g16['fence_armed'], trace_equal = e16['trace'] == g16['trace'], params_equal = bool(torch.equal(e16['params'], g16['params'])),
		exchange_bytes = e16['exchange'], exchange_bytes_fp32 = ref16['exchange'],
		first_loss_equal_to_fp32_exchange = e16['trace'][0][0] == ref16['trace'][0][0],
		first_update_at = (e16['after1'][0], ref16['after1'][0]),
		first_update_rel_to_fp32_exchange = float((e16['after1'][1] - ref16['after1'][1]).norm() / (ref16['after1'][1] - ref16['p0']).norm()),
		grad_norms = [t[2] for t in e16['trace']], grad_norms_graph = [t[2] for t in g16['trace']], grad_norms_fp32_exchange = [t[2] for t in ref16['trace']])
except Exception as e:
	import traceback
	traceback.print_exc()
	res['error'] = f'{type(e).__name__}: {e}'
dist.destroy_process_group()
print(json.dumps(res), flush = True)
