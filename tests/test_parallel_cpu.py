"""World-size-2 gloo tests (CPU) of the data-parallel engine's host logic: flat arenas, bucketing, readiness-ordered launches,
mean-reduction, broadcast of the initial replica.  The kernels themselves are covered by the -m gpu tests."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	return port


class Toy(nn.Module):
	def __init__(self):
		super().__init__()
		self.a = nn.Conv1d(8, 16, 3, bias = False)
		self.bn = nn.BatchNorm1d(16)
		self.b = nn.Conv1d(16, 4, 1)


def _worker(rank, world, port, out):
	os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
	dist.init_process_group('gloo', rank = rank, world_size = world)
	from convasr_amd.parallel import DataParallelEngine
	from convasr_amd.functional import _deliver
	torch.manual_seed(100 + rank)  # different initial replicas: rank 0's must win
	model = Toy()
	engine = DataParallelEngine(model, bucket_bytes = 1024, fold_mean = False)  # eager mean: the assertions below read p.grad directly
	flat = engine.flat
	assert len(engine.buckets) >= 2
	w0 = flat.data.clone()
	gathered = [torch.empty_like(w0) for _ in range(world)]
	dist.all_gather(gathered, w0)
	assert all(torch.equal(g, gathered[0]) for g in gathered), 'replicas differ after broadcast'
	assert model.a.weight.data_ptr() == flat.data.data_ptr(), 'parameters must be views of the arena'

	launched = []
	orig = engine._launch
	engine._launch = lambda bi, events = (): (launched.append(bi), orig(bi, events))[1]
	params = list(model.parameters())
	expect = sum(range(1, world + 1)) / world
	# step 1 -- "backward": gradients become final in reverse layer order; every rank contributes rank + 1
	for p in reversed(params):
		_deliver([p], lambda outs, acc: outs[0].fill_(float(rank + 1)) if not acc else outs[0].add_(float(rank + 1)))
	assert launched == list(reversed(range(len(engine.buckets)))), ('buckets must launch decoder-side first, as they complete', launched)
	engine.finish_gradient_sync()
	for p in params:
		assert torch.allclose(p.grad, torch.full_like(p, expect)), (rank, p.grad.flatten()[:4])
	# step 2 -- one parameter gets no gradient: its bucket is flushed (as zeros) by finish_gradient_sync
	flat.zero_grad()
	del launched[:]
	for p in reversed(params):
		if p is not model.bn.bias:
			_deliver([p], lambda outs, acc: outs[0].fill_(float(rank + 1)) if not acc else outs[0].add_(float(rank + 1)))
	engine.finish_gradient_sync()
	assert sorted(launched) == list(range(len(engine.buckets))), launched
	for p in params:
		want = 0.0 if p is model.bn.bias else expect
		assert torch.allclose(p.grad, torch.full_like(p, want)), (rank, want, p.grad.flatten()[:4])

	# step 3 -- gradient accumulation: the first backward of the group runs under no_sync(), the second launches the buckets
	flat.zero_grad()
	del launched[:]
	with engine.no_sync():
		for p in reversed(params):
			_deliver([p], lambda outs, acc: outs[0].fill_(float(rank + 1)) if not acc else outs[0].add_(float(rank + 1)))
	assert launched == []
	for p in reversed(params):
		_deliver([p], lambda outs, acc: outs[0].fill_(float(rank + 1)) if not acc else outs[0].add_(float(rank + 1)))
	engine.finish_gradient_sync()
	for p in params:
		assert torch.allclose(p.grad, torch.full_like(p, 2 * expect)), (rank, p.grad.flatten()[:4])
	dist.barrier()
	dist.destroy_process_group()
	out.put((rank, 'ok'))


@pytest.mark.parametrize('world', [2, 8])
def test_data_parallel_engine_gloo(world):
	ctx = mp.get_context('spawn')
	out = ctx.Queue()
	port = _free_port()
	procs = [ctx.Process(target = _worker, args = (r, world, port, out)) for r in range(world)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(180)
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	assert sorted(out.get(timeout = 5)[0] for _ in range(world)) == list(range(world))


def test_bucket_grading_and_hook_lifetime():
	"""Buckets are contiguous arena ranges in parameter order, graded 4, 8, 16, ... up to the limit from the START of the arena (the
	part of the exchange nothing overlaps), and an engine that is closed or dropped leaves nothing behind in the backward pass's hook
	registry."""
	import gc
	from convasr_amd.parallel import DataParallelEngine
	from convasr_amd import functional as Fn

	class Wide(nn.Module):
		def __init__(self):
			super().__init__()
			self.layers = nn.ModuleList(nn.Conv1d(64, 64, 1, bias = False) for _ in range(40))  # 16 KiB each
	before = set(Fn.after_long_launch_hooks)
	eng = DataParallelEngine(Wide(), bucket_bytes = 128 << 10, first_bucket_bytes = 16 << 10)
	sizes = [(b['hi'] - b['lo']) * 4 for b in eng.buckets]
	assert sizes[:4] == [16 << 10, 32 << 10, 64 << 10, 128 << 10] and max(sizes) == 128 << 10, sizes
	assert all(a['hi'] <= b['lo'] for a, b in zip(eng.buckets, eng.buckets[1:])) and eng.buckets[0]['lo'] == 0
	assert sum(len(b['params']) for b in eng.buckets) == 40
	assert len(set(Fn.after_long_launch_hooks) - before) == 1
	eng.close()
	assert set(Fn.after_long_launch_hooks) == before
	eng2 = DataParallelEngine(Wide())
	assert len(set(Fn.after_long_launch_hooks) - before) == 1
	del eng2
	gc.collect()
	Fn._after_long_launch()  # a stale entry, had one survived, removes itself here
	assert set(Fn.after_long_launch_hooks) == before


def test_predicted_exposed_comm_model():
	"""parallel.predict_exposed_comm (the prediction a first multi-GPU curve can be checked against): buckets complete from
	the end of the arena, collectives are serial on the communication stream, exposure is what runs past the end of the backward pass."""
	import convasr_amd as ca
	from convasr_amd.parallel import DataParallelEngine, predict_exposed_comm
	model = ca.models.Wav2Letter(64, [38], base_width = 32).train()
	eng = DataParallelEngine(model)
	try:
		p1 = eng.predict(1, 16.0)
		assert p1['exposed_comm_ms'] == 0.0 and p1['comm_ms_total'] == 0.0 and p1['predicted_scaling'] is None
		p = {n: eng.predict(n, 16.0) for n in (2, 4, 8)}
		assert p[2]['comm_ms_total'] > p[4]['comm_ms_total'] > p[8]['comm_ms_total'] > 0  # S / N per link: more peers, more links
		for n, q in p.items():
			rows = q['per_bucket']
			assert len(rows) == len(eng.buckets) and all(a['ready_ms'] <= b['ready_ms'] for a, b in zip(rows, rows[1:]))
			assert all(r['start_ms'] >= r['ready_ms'] and abs(r['end_ms'] - r['start_ms'] - r['comm_ms']) < 2e-3 for r in rows)
			assert all(a['end_ms'] <= b['start_ms'] + 1e-9 for a, b in zip(rows, rows[1:]))  # serial on the communication stream
			assert abs(rows[-1]['ready_ms'] - q['backward_end_ms']) < 1e-3  # the first layers' bucket completes when backward ends
			assert q['exposed_comm_ms'] >= rows[-1]['comm_ms'] - 1e-3 and 1.0 < q['predicted_scaling'] <= n
		slow = eng.predict(8, 16.0, efficiency = 0.01)
		assert slow['exposed_comm_ms'] > p[8]['exposed_comm_ms'] and slow['predicted_scaling'] < p[8]['predicted_scaling']
	finally:
		eng.close()


def test_exchange_type_capturability_and_halved_prediction():
	"""Round 6: what the gradient buckets travel as (DataParallelEngine(grad_comm_dtype)) and whether a step may be captured, decided without a
	GPU: CPU arenas always exchange the fp32 arena in place and a gloo engine is never captured; 'auto' follows the model's compute type;
	the prediction with 2 bytes per element moves half the bytes through every collective."""
	import convasr_amd as ca
	from convasr_amd.parallel import DataParallelEngine
	model = ca.models.Wav2Letter(64, [38], base_width = 32).train()
	eng = DataParallelEngine(model)
	try:
		assert eng.grad_comm_dtype == 'auto' and eng.comm_dtype() is None  # a CPU arena: no 16-bit send buffers
		assert eng.capturable  # (no collectives: nothing stands in a capture's way)
		full = eng.exchange_bytes()
		assert full == sum((b['hi'] - b['lo']) * 4 for b in eng.buckets)
		for spec, want in ((None, None), ('f32', None), ('f16', torch.float16), ('bf16', torch.bfloat16), (torch.float16, torch.float16)):
			assert DataParallelEngine(model, flat = eng.flat, grad_comm_dtype = spec).grad_comm_dtype == want
		a, b = eng.predict(8, 16.0, bytes_per_element = 4), eng.predict(8, 16.0, bytes_per_element = 2)
		lat = 8 * 40e-3  # eight-ish collectives' fixed latency does not halve
		assert b['bytes_per_element'] == 2 and 0.5 * a['comm_ms_total'] <= b['comm_ms_total'] <= 0.5 * a['comm_ms_total'] + len(eng.buckets) * 0.04 + 1e-6
		assert b['exposed_comm_ms'] <= a['exposed_comm_ms']
	finally:
		eng.close()
	from convasr_amd import rccl
	assert rccl._NCCL_DTYPES[torch.float16] == 6 and rccl._NCCL_DTYPES[torch.float32] == 7 and rccl._NCCL_DTYPES[torch.bfloat16] == 9  # rccl.h: ncclDataType_t
	import ctypes
	assert ctypes.sizeof(rccl._UniqueId) == 128


def test_flat_parameters_views_and_state_dict_roundtrip():
	from convasr_amd.train import FlatParameters
	model = Toy()
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	flat = FlatParameters(model)
	for k, v in model.state_dict().items():
		assert torch.equal(v, sd[k]), k
	assert flat.numel % FlatParameters.ALIGN == 0
	with torch.no_grad():
		flat.data.add_(1.0)
	assert torch.allclose(model.a.weight, sd['a.weight'] + 1.0)
	model.load_state_dict(sd)  # load_state_dict copies in place: the views survive
	assert model.a.weight.data_ptr() == flat.data.data_ptr()
	assert torch.equal(model.b.bias, sd['b.bias'])


def test_lr_schedules_host_arithmetic():
	"""optimizers.py:9-63 mirrors: schedules only touch param_groups[...]['lr'] (no device work: importable without a GPU)."""
	import importlib.util, sys, types
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	# load convasr_amd/optimizers.py without importing the package (the package refuses to import without its HIP library)
	pkg = types.ModuleType('_cv_stub'); pkg.__path__ = [os.path.join(root, 'convasr_amd')]
	sys.modules['_cv_stub'] = pkg
	for name in ('ops', 'functional'):
		sys.modules[f'_cv_stub.{name}'] = types.ModuleType(f'_cv_stub.{name}')
	spec = importlib.util.spec_from_file_location('_cv_stub.optimizers', os.path.join(root, 'convasr_amd', 'optimizers.py'))
	opt_mod = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(opt_mod)

	class FakeOpt:
		def __init__(self):
			self.defaults = dict(lr = 0.1)
			self.param_groups = [dict(lr = 0.1), dict(lr = 0.2)]
	o = FakeOpt()
	s = opt_mod.MultiStepLR(o, gamma = 0.5, milestones = [10, 20])
	for step, want in [(0, 0.1), (9, 0.1), (10, 0.05), (19, 0.05), (20, 0.025), (1000, 0.025)]:
		s.step(step)
		assert abs(o.param_groups[0]['lr'] - want) < 1e-12 and abs(o.param_groups[1]['lr'] - 2 * want) < 1e-12
	o = FakeOpt()
	s = opt_mod.PolynomialDecayLR(o, decay_steps = 100, power = 2.0, begin_decay_at = 50, end_lr = 0.01, warmup_steps = 10)
	s.step(5); assert abs(o.param_groups[0]['lr'] - 0.05) < 1e-12
	s.step(20); assert abs(o.param_groups[0]['lr'] - 0.1) < 1e-12
	s.step(100); assert abs(o.param_groups[0]['lr'] - (0.01 + 0.09 * 0.25)) < 1e-12
	s.step(500); assert abs(o.param_groups[0]['lr'] - 0.01) < 1e-12
	o = FakeOpt()
	opt_mod.NoopLR(o).step(7); assert o.param_groups[0]['lr'] == 0.1
	o.param_groups[0]['lr'] = 3.0
	opt_mod.reset_options(o); assert o.param_groups[0]['lr'] == 0.1


def _run_bench(extra_env, *argv):
	import subprocess, sys
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	env = dict(os.environ, **extra_env)
	env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
	return subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *argv], env = env, stdout = subprocess.PIPE, stderr = subprocess.PIPE, text = True, timeout = 300)


def test_bench_launches_its_own_ranks_world2():
	"""`python bench.py --gpus 2` (no WORLD_SIZE): the parent starts torch.distributed.run as a child, relays rank 0's single JSON
	line and returns the child's code (train.py:1057-1073 spawns its ranks the same way).  --launcher-dry-run keeps the ranks on
	CPU tensors over gloo so the launch path itself runs here."""
	import json
	r = _run_bench({}, '--gpus', '2', '--launcher-dry-run', '--steps', '4', '--warmup', '1')
	assert r.returncode == 0, r.stderr[-2000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
	assert len(lines) == 1, lines
	out = json.loads(lines[0])
	assert out['n_gpus'] == 2 and out['steps'] == 4 and out['warmup'] == 1
	assert out['dist'] == dict(backend = 'gloo', world_size = 2, launcher = True)


def test_bench_launcher_propagates_a_rank_failure():
	r = _run_bench(dict(CONVASR_DRY_RUN_FAIL_RANK = '1'), '--gpus', '2', '--launcher-dry-run')
	assert r.returncode != 0


def test_bench_launcher_world8_dry_run():
	r = _run_bench({}, '--gpus', '8', '--launcher-dry-run')
	assert r.returncode == 0, r.stderr[-2000:]
	line = json.loads(r.stdout.strip().splitlines()[-1])
	assert line['n_gpus'] == 8 and line['value'] == 36.0 and line['dist']['world_size'] == 8


def test_bench_launcher_ends_a_hung_child_tree():
	"""A rank that never reaches its collective: the parent gives up after CONVASR_LAUNCH_TIMEOUT, ends the whole child tree
	(its own session) and exits 124 -- no orphan keeps the port or a GPU."""
	import subprocess, time
	t0 = time.time()
	r = _run_bench(dict(CONVASR_DRY_RUN_HANG_RANK = '1', CONVASR_LAUNCH_TIMEOUT = '20'), '--gpus', '2', '--launcher-dry-run')
	assert r.returncode == 124 and 'ending the child tree' in r.stderr, (r.returncode, r.stderr[-1500:])
	assert time.time() - t0 < 90
	time.sleep(1.0)
	left = subprocess.run(['pgrep', '-f', 'launcher-dry-run'], stdout = subprocess.PIPE, text = True).stdout.split()
	assert not left, left


def test_bench_rank_preflight_and_flop_count():
	"""pin_to_gpu_numa_node is best effort (no KFD topology in the CPU container: it must say so, not raise); conv_stack_flops
	reproduces SURVEY 8(d)'s figures from the module tree."""
	import bench
	import convasr_amd as ca
	info = bench.pin_to_gpu_numa_node(0)
	assert isinstance(info, dict) and 'pinned' in info and (info['pinned'] or 'reason' in info)
	assert bench._cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	fwd, bwd = bench.conv_stack_flops(ca.models.Wav2Letter(64, [38], frontend = fe), 64, 15 * 16000)
	assert abs(fwd / 1e9 - 6399) < 2 and abs((fwd + bwd) / (64 * 15) / 1e9 - 19.98) < 0.01, (fwd, bwd)
	fwd, bwd = bench.conv_stack_flops(ca.models.JasperNetLarge(64, [38], frontend = fe), 32, 20 * 16000)
	assert abs(fwd / 1e9 - 17751) < 20, fwd


def test_bench_launcher_refuses_more_ranks_than_gpus():
	r = _run_bench(dict(CONVASR_SHARE_GPU = '0'), '--gpus', '2')  # no GPU in the CPU container: a clear message, not a hang
	assert r.returncode == 2 and 'GPU(s) are visible' in r.stderr
