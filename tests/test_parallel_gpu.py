"""Two data-parallel ranks running the REAL kernels (both on cuda:0, collectives over gloo, which stages CUDA tensors through
the host): the bucketed all-reduce hooks of DataParallelEngine inside a real backward, replicas staying identical, and the
averaged update equal to a single process that computes the two ranks' gradients one after the other.  (RCCL itself needs one
GPU per rank: the driver's multi-GPU bench is the first place it runs with world > 1; bench.py's CONVASR_FORCE_DIST=1 mode
covers it with world = 1.)"""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	return port


def _make(ca, d):
	torch.manual_seed(0)
	return ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.0], out_width_factors_large = [2, 2], residual = False, repeat = 1, check_time_dim_padded = False, temporal_mask = True).to(d).train()


def _batch(rank, d):
	g = torch.Generator().manual_seed(50 + rank)
	x = torch.randn(3, 64, 48, generator = g).to(d)
	xlen = torch.tensor([1.0, 0.8, 0.6]).to(d)
	y = torch.randint(0, 37, (3, 1, 5), generator = g).to(d)
	ylen = torch.tensor([[5], [4], [3]]).to(d)
	return x, xlen, y, ylen


def _worker(rank, world, port, out_dir, side = False):
	os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
	dist.init_process_group('gloo', rank = rank, world_size = world)
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.cuda.set_device(d)
	model = _make(ca, d)
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	engine = ca.parallel.DataParallelEngine(model, device = d, bucket_bytes = 64 << 10)
	assert len(engine.buckets) >= 2
	if side:
		ca.functional.enable_side_stream_wgrad(d)
	batch = _batch(rank, d)
	losses = []
	# (second iteration through train.GraphedTrainStep: an engine that runs collectives is never captured -- the call is train_step's)
	stepper = ca.train.GraphedTrainStep(engine, opt, warmup = 1, world_size = world, sync_metrics = True)
	assert not stepper.enabled
	for it in range(2):
		res = ca.train.train_step(engine, opt, *batch, world_size = world, iteration = it, sync_metrics = True) if it == 0 else stepper(*batch, iteration = it)
		assert not bool(res['skipped'])
		losses.append(float(res['loss_cur']))
	assert stepper.eager_steps == 1 and stepper.captures == 0
	torch.cuda.synchronize()
	torch.save(dict(params = flat.data.cpu(), losses = losses), os.path.join(out_dir, f'rank{rank}{"_side" if side else ""}.pt'))
	dist.barrier()
	dist.destroy_process_group()


def test_two_ranks_real_kernels_match_sequential_average():
	import convasr_amd as ca
	world = 2
	with tempfile.TemporaryDirectory() as out_dir:
		mp.spawn(_worker, args = (world, _free_port(), out_dir), nprocs = world, join = True)
		got = [torch.load(os.path.join(out_dir, f'rank{r}.pt')) for r in range(world)]
	assert torch.equal(got[0]['params'], got[1]['params']), 'replicas diverged'
	assert got[0]['losses'] == got[1]['losses'], 'the all-reduced metric must be the same number on every rank'

	# single process: per-rank replicas of the batch-norm buffers, shared parameters, gradients averaged by hand
	d = torch.device('cuda:0')
	models = [_make(ca, d) for _ in range(world)]
	flats = [ca.train.FlatParameters(m) for m in models]
	for m, f in zip(models, flats):
		m._convasr_flat = f
	opt = ca.train.SGD(flats[0], lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	for it in range(2):
		grads = []
		for r in range(world):
			flats[r].data.copy_(flats[0].data)
			ca.functional.bump_param_epoch()
			x, xlen, y, ylen = _batch(r, d)
			out = models[r](x, xlen, y = y, ylen = ylen)
			((out['loss'] * ylen[:, 0]).mean()).backward()
			flats[r].finalize_grads()
			grads.append(flats[r].grad.clone())
			flats[r].zero_grad()
		flats[0].grad.copy_(sum(grads) / world)
		for p in flats[0].params:
			p._convasr_fresh = False
		flats[0].clip_grad_norm_(100.0)
		opt.step()
		opt.zero_grad()
	ref = flats[0].data.cpu()
	err = (got[0]['params'] - ref).abs().max().item()
	assert err <= 2e-6 + 1e-5 * ref.abs().max().item(), err


def test_two_ranks_side_stream_wgrad_equals_single_stream_bitwise():
	"""enable_side_stream_wgrad under the data-parallel engine: a bucket mixes weight gradients produced on the side stream with
	dgamma / dbeta produced on the main stream; the engine issues each bucket's all-reduce from a stream that waits for BOTH
	producers (parallel.DataParallelEngine._launch), so the replicas end bit-identical to the single-stream run."""
	world = 2
	with tempfile.TemporaryDirectory() as out_dir:
		mp.spawn(_worker, args = (world, _free_port(), out_dir, False), nprocs = world, join = True)
		mp.spawn(_worker, args = (world, _free_port(), out_dir, True), nprocs = world, join = True)
		plain = [torch.load(os.path.join(out_dir, f'rank{r}.pt')) for r in range(world)]
		side = [torch.load(os.path.join(out_dir, f'rank{r}_side.pt')) for r in range(world)]
	assert torch.equal(side[0]['params'], side[1]['params']), 'replicas diverged with the side stream on'
	assert torch.equal(side[0]['params'], plain[0]['params']), float((side[0]['params'] - plain[0]['params']).abs().max())
	assert side[0]['losses'] == plain[0]['losses']


def _worker_fp16(rank, world, port, out_dir):
	os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
	dist.init_process_group('gloo', rank = rank, world_size = world)
	import convasr_amd as ca
	d = torch.device('cuda:0')
	torch.cuda.set_device(d)
	model = _make(ca, d)
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	engine, opt = ca.models.distributed_data_parallel_and_autocast(model, 0, opt, opt_level = 'O2')
	assert isinstance(engine, ca.parallel.DataParallelEngine) and model.compute_dtype == torch.float16 and flat.loss_scaler is not None
	flat.loss_scaler = ca.train.LossScaler(d, init_scale = 2.0 ** 20)
	x, xlen, y, ylen = _batch(rank, d)
	history = []
	poison = [False]
	launch = engine._launch

	def launch_with_private_overflow(bi, events = ()):
		# rank 1, iteration 6: ITS local gradient holds an inf (as if its own backward had overflowed fp16) just before bucket 0 is
		# all-reduced -- the verdict is read from the all-reduced gradient, so BOTH ranks must skip that step
		if poison[0] and bi == 0:
			comm = engine._comm(d)
			for ev in events:
				comm.wait_event(ev)
			with torch.cuda.stream(comm):
				flat.grad[engine.buckets[0]['lo']:engine.buckets[0]['lo'] + 1].fill_(float('inf'))
		return launch(bi, events)
	engine._launch = launch_with_private_overflow
	for it in range(10):
		poison[0] = it == 6 and rank == 1
		res = ca.train.train_step(engine, opt, x, xlen, y, ylen, world_size = world, iteration = it, sync_metrics = True)
		history.append((flat.loss_scaler.state_dict()['loss_scale'], bool(torch.isfinite(res['grad_norm']))))
	torch.cuda.synchronize()
	torch.save(dict(params = flat.data.cpu(), history = history, scaler = flat.loss_scaler.current.cpu()), os.path.join(out_dir, f'fp16_rank{rank}.pt'))
	dist.barrier()
	dist.destroy_process_group()


def test_two_ranks_fp16_loss_scaler_skips_on_every_rank_or_on_none():
	"""fp16 + dynamic loss scaling under the data-parallel engine (models.distributed_data_parallel_and_autocast(opt_level = 'O2')): the
	overflow verdict is the non-finite norm of the ALL-REDUCED gradient, so it is the same on every rank: the start-up overflows (scale
	2^20 here) and an overflow only rank 1's batch causes are skipped by both; replicas, scale and counters end identical."""
	world = 2
	with tempfile.TemporaryDirectory() as out_dir:
		mp.spawn(_worker_fp16, args = (world, _free_port(), out_dir), nprocs = world, join = True)
		got = [torch.load(os.path.join(out_dir, f'fp16_rank{r}.pt')) for r in range(world)]
	assert torch.equal(got[0]['params'], got[1]['params']), 'replicas diverged'
	assert got[0]['history'] == got[1]['history'] and torch.equal(got[0]['scaler'], got[1]['scaler'])
	finite = [h[1] for h in got[0]['history']]
	print('fp16 two ranks:', got[0]['history'])
	assert not finite[0] and any(finite) and not finite[6], got[0]['history']  # start-up overflow, clean steps, and rank 1's private overflow skipped by both
	assert bool(torch.isfinite(got[0]['params']).all())


def test_rccl_world1_data_parallel_step_replays_bitwise_from_graphs_with_fp32_and_fp16_exchange():
	"""A one-rank RCCL process group (a child process): the data-parallel step with its bucket all-reduces captured on the communication
	stream replays bit for bit like the eager step (two batch shapes, dropout, fp32 exchange); the same under apex O2 with the gradients
	exchanged as fp16 (half the bytes, grad_comm_dtype 'auto'), whose result stays within fp16 rounding of the fp32 exchange's."""
	import json
	import subprocess
	import sys
	here = os.path.dirname(os.path.abspath(__file__))
	r = subprocess.run([sys.executable, os.path.join(here, '_dp_rccl_world1.py'), str(_free_port())], stdout = subprocess.PIPE, stderr = subprocess.PIPE, text = True, timeout = 900)
	lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
	assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
	res = json.loads(lines[-1])
	assert 'error' not in res, (res, r.stderr[-4000:])
	a, b = res['fp32_exchange'], res['fp16_exchange']
	assert a['captures'] == 2 and a['replays'] >= 7 and a['trace_equal'] and a['params_equal'], a
	assert b['comm_dtype'] == 'torch.float16' and b['captures'] == 2 and b['trace_equal'] and b['params_equal'], b
	assert b['exchange_bytes'] * 2 == b['exchange_bytes_fp32'] and b['first_loss_equal_to_fp32_exchange'], b
	# what the captures hold: RCCL's all-reduces are kernel nodes like everything else of the step (a memset / memcpy node would arm the eager / replay fence)
	for leg in (a, b):
		assert leg['node_kinds'] and all(k is not None and k.get('kernel', 0) >= 100 for k in leg['node_kinds']), leg['node_kinds']
		assert leg['fence_armed'] == any(set(k) != {'kernel'} for k in leg['node_kinds']), leg
	# the first applied update with the gradients exchanged as fp16 against the same update with the fp32 exchange: apart by fp16's rounding of
	# the gradient (2^-11 per element), relative to the update itself.  (A gradient of the scaled loss that fits fp32 but not fp16 overflows in the
	# 16-bit send buffer -- as it would in apex O2's fp16 gradients -- and costs one more skipped start-up step: compared when both runs applied the same step.)
	if b['first_update_at'][0] == b['first_update_at'][1]:
		assert b['first_update_rel_to_fp32_exchange'] <= 2e-3, b
	print('rccl world 1:', res)
