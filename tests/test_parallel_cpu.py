"""World-size-2 gloo tests (CPU) of the data-parallel engine's host logic: flat arenas, bucketing, readiness-ordered launches,
mean-reduction, broadcast of the initial replica.  The kernels themselves are covered by the -m gpu tests."""
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


def test_data_parallel_engine_gloo_world2():
	ctx = mp.get_context('spawn')
	out = ctx.Queue()
	port = _free_port()
	procs = [ctx.Process(target = _worker, args = (r, 2, port, out)) for r in range(2)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(120)
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	assert sorted(out.get(timeout = 5)[0] for _ in range(2)) == [0, 1]


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


def test_bench_launcher_refuses_more_ranks_than_gpus():
	r = _run_bench(dict(CONVASR_SHARE_GPU = '0'), '--gpus', '2')  # no GPU in the CPU container: a clear message, not a hang
	assert r.returncode == 2 and 'GPU(s) are visible' in r.stderr
