"""bench.py -- BASELINE.json's headline metric on the MI355X-native path.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 without WORLD_SIZE in the environment: this process touches no GPU and starts its own N ranks as a CHILD
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, the role of train.py:1057-1073's mp.spawn), relays rank 0's
JSON line and exits with the child's return code.  Under torch.distributed.run (WORLD_SIZE set) it is one rank of that job.

A step = one full training iteration of the hot path on one synthetic batch per GPU (train.py:745-783 of the reference):
logmel frontend -> instance norm -> Wav2Letter full (18 x Conv1d+BN+hardtanh+dropout+mask, 1x1 decoder) -> log-softmax ->
CTC loss -> backward (dgrad / wgrad / BN / CTC) -> gradient all-reduce (N > 1) -> clip_grad_norm_ -> SGD.  Workload =
BASELINE configs[2]/[3]: 64 utterances x 15 s of 16 kHz audio per GPU, bf16 MFMA convolutions with fp32 accumulation and
fp32 master weights, dropout 0.2 (--dtype f16: fp16 storage + MFMA under apex's dynamic loss scaling, the arithmetic BASELINE
configs[4] names; --dtype f32: the exact-fp32 parity path).  Inputs are resident in HBM before the timed region.  Prints ONE JSON
line on rank 0; its `parity` object is the second half of BASELINE's metric: the CTC loss of the GPU paths (fp32, bf16, fp16)
relative to the CPU oracle on the sample the cpu_baseline leg runs anyway.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL / cross-process GPU buffers need it on this driver (already exported on the pool)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE, SECS, BATCH, TARGET_LEN = 16000, 15, 64, 150
FLOP_PER_AUDIO_S_FWD_BWD = 19.98e9  # conv stack, 2*MAC, fwd + dgrad + wgrad
PEAK_BF16_DENSE = 2.5e15  # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MAIN_KERNEL_SYMBOLS = dict(bf16 = ('conv1d_igemm_v2s_kernel<unsigned short, unsigned short', 'conv1d_igemm_v2s_kernelIttLi'), f16 = ('conv1d_igemm_v2s_kernel<_Float16, _Float16', 'conv1d_igemm_v2s_kernel<__half, __half', 'conv1d_igemm_v2s_kernelIDF16_DF16_Li'))  # demangled or (where the profiler's demangler does not know _Float16) mangled;  # <H, H, false> (plain) and <H, H, true> (dgrad + fused BN-backward epilogue); NOT <H, float, false>, the decoder head


def parse_args(argv = None):
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type = int, default = 1)
	ap.add_argument('--steps', type = int, default = 10)
	ap.add_argument('--warmup', type = int, default = 3)
	ap.add_argument('--dtype', default = 'bf16', choices = ['bf16', 'f16', 'f32'])
	ap.add_argument('--dropout', type = float, default = 0.2, help = 'the reference Wav2Letter default is 0.2; other values are for experiments only')
	ap.add_argument('--no-cpu-baseline', action = 'store_true')
	ap.add_argument('--no-kernel-timer', action = 'store_true')
	ap.add_argument('--no-traffic', action = 'store_true', help = 'skip the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic in this run')
	ap.add_argument('--side-stream', action = 'store_true', help = 'run wgrad on a second HIP stream (+1.5-2 % step rate; off by default so that the per-kernel HIP-event durations of the roofline leg are not inflated by overlap)')
	ap.add_argument('--launcher-dry-run', action = 'store_true', help = 'test hook: ranks only rendezvous (gloo, CPU tensors) and rank 0 prints a line; exercises the self-launch path without a GPU')
	return ap.parse_args(argv)


def _free_port():
	s = socket.socket()
	s.bind(('127.0.0.1', 0))
	port = s.getsockname()[1]
	s.close()
	return port


def _last_json_line(text):
	for ln in reversed(text.splitlines()):
		ln = ln.strip()
		if ln.startswith('{') and ln.endswith('}'):
			try:
				obj = json.loads(ln)
			except ValueError:
				continue
			if 'metric' in obj:
				return ln
	return None


def launch_ranks(args, argv):
	"""Parent side of `python bench.py --gpus N`: no GPU call is made here (torch.cuda.device_count() does not initialise the
	device on this image), the ranks run in a child process tree started by torch.distributed.run."""
	if not args.launcher_dry_run and os.environ.get('CONVASR_SHARE_GPU') != '1':
		import torch
		have = torch.cuda.device_count()
		if have < args.gpus:
			print(f'bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible', file = sys.stderr)
			return 2
	cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__), *argv]
	proc = subprocess.run(cmd, stdout = subprocess.PIPE, text = True)  # stderr is inherited
	line = _last_json_line(proc.stdout or '')
	if line is not None:
		print(line, flush = True)
	elif proc.stdout:
		sys.stderr.write(proc.stdout)
	if proc.returncode != 0:
		return proc.returncode
	return 0 if line is not None else 1


def synthetic_batch(device, batch = BATCH, secs = SECS, seed = 1):
	import torch
	g = torch.Generator().manual_seed(seed)
	x = torch.rand(batch, SAMPLE_RATE * secs, generator = g) * 2 - 1
	xlen = torch.ones(batch)
	y = torch.randint(0, 37, (batch, 1, 10 * secs), generator = g)
	ylen = torch.full((batch, 1), 10 * secs, dtype = torch.long)
	return tuple(t.to(device) for t in (x, xlen, y, ylen))


def cpu_baseline(secs = SECS, batch = 4, iters = 3, keep = None):
	"""The oracle (kind 'port': plain-torch CPU restatement of the reference's path, pinned to the reference by
	tests/golden) timed on this host's cores on a bounded sample of the same workload: `batch` x 15 s
	utterances, fwd + CTC + bwd + clip + SGD, 1 warm-up + `iters` timed iterations, mean and best reported.  Threads: BASELINE.md
	prescribes os.cpu_count(), but torch's CPU conv / BN kernels oversubscribe badly on a 256-thread host (measured on the GPU box,
	2 x 15 s: 8 threads 69, 16 threads 118, 32 threads 90, 64 threads 43, 128 threads 20, 256 threads 0.8 audio-s/s:
	profiles/README.md), so the default is min(os.cpu_count(), 16), the fastest setting; CONVASR_CPU_THREADS overrides it and the
	count actually used is in the result.  keep (a dict): receives the warm-up iteration's inputs, initial parameters and
	per-utterance CTC losses -- the reference values of the `parity` leg."""
	import torch
	from oracle import convasr_oracle as O
	cores = int(os.environ.get('CONVASR_CPU_THREADS', 0)) or min(os.cpu_count() or 1, 16)
	torch.set_num_threads(cores)
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	fe = O.frontend_config()
	sd = O.init_state_dict(plan, seed = 1, frontend = fe)
	x, xlen, y, ylen = synthetic_batch('cpu', batch = batch, secs = secs)
	bufs = {}
	times = []
	if keep is not None:
		keep.update(sd = {k: v.clone() for k, v in sd.items()}, batch = (x, xlen, y, ylen))
	for it in range(1 + iters):
		t0 = time.perf_counter()
		r = O.train_step(sd, plan, x, xlen, y, ylen, frontend = fe, momentum_buffers = bufs)
		times.append(time.perf_counter() - t0)
		if it == 0 and keep is not None:
			keep.update(loss_vec = r['loss_vec'].clone(), loss = float(r['loss']))
	timed = times[1:]
	mean, best = sum(timed) / len(timed), min(timed)
	return dict(value = round(batch * secs / mean, 2), best = round(batch * secs / best, 2), unit = 'audio-seconds/sec', cores = torch.get_num_threads(), host_cpus = os.cpu_count(), kind = 'port', sample = f'{batch}x{secs}s utterances, Wav2Letter full fp32, fwd+CTC+bwd+clip+SGD, mean of {iters} timed iterations after 1 warm-up ({mean:.2f} s/step mean, {best:.2f} s/step best)')


def gpu_parity(ref, device):
	"""BASELINE's "CTC loss rel-err vs ref": the per-utterance CTC losses of the MI355X path, for every compute type, against the CPU
	oracle's on the cpu_baseline leg's sample (same inputs, same initial parameters, train-mode batch statistics, dropout 0 -- the
	oracle has no dropout; the headline throughput runs with 0.2).  Relative error = max over utterances."""
	import torch
	import convasr_amd as ca
	x, xlen, y, ylen = (t.to(device) for t in ref['batch'])
	out = {}
	for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16), ('f16', torch.float16)):
		fe = ca.models.LogFilterBankFrontend(64, SAMPLE_RATE, 0.02, 0.01, 'hann_window')
		model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.0, check_time_dim_padded = False, compute_dtype = dt)
		missing = model.load_state_dict(ref['sd'], strict = False)
		assert not missing.missing_keys, missing
		model.to(device).train()
		with torch.no_grad():
			loss = model(x, xlen, y = y, ylen = ylen)['loss'].float().cpu()
		out[name] = float(((loss - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
		del model
	return dict(ctc_loss_rel_err = {k: float(f'{v:.3e}') for k, v in out.items()}, north_star_bound = 1e-4, reference = 'oracle (fp32 CPU restatement of the reference path, pinned to the reference by tests/golden)', sample = f'{x.shape[0]}x{x.shape[1] // SAMPLE_RATE}s utterances of the cpu_baseline leg, same initial parameters, train-mode forward, dropout 0; per-utterance CTC loss, max relative error', note = 'f32 is the parity path (within north_star\'s 1e-4); bf16 (8 significant bits of storage) and f16 (11) buy their throughput at the error shown: the deviation is the storage type\'s own (tests/test_round2_gpu.py: a CPU restatement with the same storage type deviates alike)')


def measure_traffic(args, sequence, steps):
	"""roofline.traffic measured in THIS run: two child `rocprofv3 --kernel-trace --pmc <counter>` passes (FETCH_SIZE and WRITE_SIZE
	separately: they do not fit one pass on gfx950) over `bench.py --steps 1 --warmup 1`, HBM-side bytes per launch of the dominant
	kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE reads half of a wide streaming read on gfx950: MI355X_MICROARCH.md,
	HBM).  The children are started as ordinary subprocesses (python itself after `--`); this process is idle meanwhile.
	sequence: the parent's own launch order of one training step ((family, symbol class) per launch, _lib.KernelTimer.sequence): the
	dispatches of the kernel symbol are matched to it in order, and only those booked under roofline.achieved are averaged (the
	decoder's memory-bound dgrad runs the same symbol and is left out of both)."""
	import csv
	import glob
	import shutil
	import tempfile
	exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
	if not os.path.exists(exe):
		return None, 'rocprofv3 not found'
	symbol = MAIN_KERNEL_SYMBOLS[args.dtype]
	per_step = [fam for fam, sym in sequence[:len(sequence) // max(steps, 1)] if sym == 'v2s16']  # the symbol's dispatches of one step, in order
	counted = [not fam.startswith('hbm:') for fam in per_step]
	out = {}
	env = dict(os.environ, TMPDIR = '/tmp')
	for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
		d = tempfile.mkdtemp(prefix = f'convasr_pmc_{counter}_', dir = '/tmp')
		try:
			cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__), '--steps', '1', '--warmup', '1', '--dtype', args.dtype, '--dropout', str(args.dropout), '--no-cpu-baseline', '--no-kernel-timer', '--no-traffic']
			r = subprocess.run(cmd, cwd = '/tmp', env = env, stdout = subprocess.PIPE, stderr = subprocess.PIPE, text = True, timeout = 600)
			files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive = True)
			if r.returncode != 0 or not files:
				return None, f'rocprofv3 --pmc {counter} failed (rc {r.returncode})'
			rows = [row for row in csv.DictReader(open(files[0])) if any(sym in row['Kernel_Name'] for sym in symbol) and row['Counter_Name'] == counter]
			rows.sort(key = lambda row: int(row.get('Dispatch_Id', 0)))
			if not rows:
				return None, f'no {symbol[0]} dispatch in the {counter} pass'
			if per_step and len(rows) % len(per_step) == 0:
				vals = [float(row['Counter_Value']) for i, row in enumerate(rows) if counted[i % len(per_step)]]
				how = f'{sum(counted)} of the {len(per_step)} dispatches of the symbol per step (the launches roofline.achieved covers)'
			else:  # the child's dispatch list does not line up with this process's launch sequence: average everything, and say so
				vals = [float(row['Counter_Value']) for row in rows]
				how = f'ALL {len(rows)} dispatches of the symbol (could not be matched to the {len(per_step)} launches per step booked here)'
			out[counter] = (sum(vals) / len(vals), len(vals), how)
		except subprocess.TimeoutExpired:
			return None, f'rocprofv3 --pmc {counter} timed out'
		finally:
			shutil.rmtree(d, ignore_errors = True)
	mb = (2 * out['FETCH_SIZE'][0] + out['WRITE_SIZE'][0]) * 1024 / 1e6
	return round(mb, 1), f'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes over bench.py --steps 1 --warmup 1, mean of {out["FETCH_SIZE"][1]} dispatches = {out["FETCH_SIZE"][2]}, MB per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024'


def dry_run_rank(args):
	"""--launcher-dry-run: rendezvous + one all-reduce on CPU tensors (gloo); rank 0 prints the line the parent relays."""
	import torch
	import torch.distributed as dist
	dist.init_process_group('gloo')
	rank, world = dist.get_rank(), dist.get_world_size()
	t = torch.tensor([float(rank + 1)])
	t0 = time.perf_counter()
	dist.all_reduce(t)
	dist.barrier()
	el = torch.tensor([time.perf_counter() - t0], dtype = torch.float64)
	dist.all_reduce(el, op = dist.ReduceOp.MAX)
	backend = dist.get_backend()
	dist.destroy_process_group()
	if int(os.environ.get('CONVASR_DRY_RUN_FAIL_RANK', '-1')) == rank:
		return 3  # test hook: the parent must propagate a rank's failure
	if rank == 0:
		assert float(t) == world * (world + 1) / 2
		print(json.dumps(dict(metric = 'launcher-dry-run', value = float(t), unit = 'sum of rank+1', n_gpus = world, steps = args.steps, warmup = args.warmup, dist = dict(backend = backend, world_size = world, launcher = os.environ.get('TORCHELASTIC_RUN_ID') is not None))), flush = True)
	return 0


def main(argv = None):
	argv = list(sys.argv[1:] if argv is None else argv)
	args = parse_args(argv)
	if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
		return launch_ranks(args, argv)
	if args.launcher_dry_run:
		return dry_run_rank(args)

	import torch

	# stdout carries exactly one line, the JSON result of rank 0: native libraries (RCCL prints a version banner through C stdio,
	# flushed at exit, i.e. AFTER a Python print) are pointed at stderr for the duration of the run
	sys.stdout.flush()
	real_stdout = os.dup(1)
	os.dup2(2, 1)

	world = int(os.environ.get('WORLD_SIZE', '1'))
	rank = int(os.environ.get('RANK', '0'))
	local_rank = int(os.environ.get('LOCAL_RANK', '0'))
	if args.gpus > 1 and world != args.gpus:
		raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE is {world}')
	# (test hooks for a 1-GPU box: CONVASR_SHARE_GPU=1 puts every rank on cuda:0, CONVASR_DIST_BACKEND=gloo replaces RCCL, which needs one GPU per rank)
	device = torch.device('cuda', 0 if os.environ.get('CONVASR_SHARE_GPU') == '1' else local_rank)
	torch.cuda.set_device(device)
	use_dist = world > 1 or os.environ.get('CONVASR_FORCE_DIST') == '1'  # the latter: single-rank RCCL smoke test of the DP path
	dist_info = None
	if use_dist:
		import torch.distributed as dist
		os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
		os.environ.setdefault('MASTER_PORT', '29511')
		os.environ.setdefault('RANK', '0')
		os.environ.setdefault('WORLD_SIZE', '1')
		backend = os.environ.get('CONVASR_DIST_BACKEND', 'nccl')
		if backend == 'nccl':
			dist.init_process_group('nccl', device_id = device)
		else:
			dist.init_process_group(backend)
		dist_info = dict(backend = dist.get_backend() + (' (RCCL)' if backend == 'nccl' else ''), world_size = dist.get_world_size(), launcher = 'torch.distributed.run' if os.environ.get('TORCHELASTIC_RUN_ID') is not None else 'env')

	import convasr_amd as ca
	from convasr_amd import _lib

	torch.manual_seed(1)
	ca.functional.manual_seed(int(os.environ.get('CONVASR_BENCH_DROPOUT_SEED', '1')) + rank)  # (the override: a measurement hook -- step time depends on the data through the chip's clock management)
	compute = dict(bf16 = torch.bfloat16, f16 = torch.float16, f32 = torch.float32)[args.dtype]
	fe = ca.models.LogFilterBankFrontend(64, SAMPLE_RATE, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = args.dropout, check_time_dim_padded = False, compute_dtype = compute).to(device).train()
	flat = ca.train.FlatParameters(model)
	model._convasr_flat = flat
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	if args.dtype == 'f16':  # apex O2: fp16 compute, fp32 masters, dynamic loss scaling from 2^16 (the start-up overflows fall into the warm-up steps)
		ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2')
		assert model.compute_dtype == torch.float16 and flat.loss_scaler is not None
	engine = ca.parallel.DataParallelEngine(model, device = device, force_collectives = use_dist) if use_dist else model
	x, xlen, y, ylen = synthetic_batch(device, seed = 1 + rank)
	if args.side_stream:
		ca.functional.enable_side_stream_wgrad(device)

	def step(i):
		return ca.train.train_step(engine, opt, x, xlen, y, ylen, world_size = world, iteration = i, sync_metrics = use_dist)

	def fence():
		if use_dist:
			dist.barrier()
		torch.cuda.synchronize()

	last = None
	for i in range(args.warmup):
		last = step(i)
	fence()
	# HIP events (on the launching stream) bracket every launch of the DOMINANT kernel inside the timed region.  The other kernel
	# families (wgrad, the HBM-bound passes, the small layers) are event-timed in a second, untimed pass of a few steps right after
	# it: an event pair costs ~5 us of stream time, and bracketing all ~110 launches of a step slowed the headline by 3.4 %
	# (18.06 vs 17.47 ms per step on one device; bracketing the dominant kernel only: ~1 %).
	main_family = 'conv1d_igemm_v2s_kernel<bf16>' if args.dtype in ('bf16', 'f16') else 'conv1d_igemm (other variants)'  # (family labels are shared by the two 16-bit types)
	overflows0 = float(flat.loss_scaler.current[7]) if flat.loss_scaler is not None else 0.0
	if not args.no_kernel_timer and rank == 0:
		_lib.timer = _lib.KernelTimer(only = [main_family, main_family + '+bn_bwd'])
	t0 = time.perf_counter()
	for i in range(args.steps):
		last = step(args.warmup + i)
	fence()
	elapsed = time.perf_counter() - t0
	kt = _lib.timer.summary() if _lib.timer is not None else {}
	sequence = list(_lib.timer.sequence) if _lib.timer is not None else []
	_lib.timer = None
	scaler_info = None
	if flat.loss_scaler is not None:
		st = flat.loss_scaler.current.tolist()
		scaler_info = dict(loss_scale = st[0], clean_steps = int(st[1]), overflowed_steps_in_timed_region = int(st[7] - overflows0), overflowed_steps_total = int(st[7]), note = 'apex dynamic loss scaling (2^16, x2 per 2000 clean steps, /2 and skip on overflow); an overflowed step skips only the optimizer update')
	steps2 = 0
	if not args.no_kernel_timer:
		steps2 = min(args.steps, 5)
		if rank == 0:
			_lib.timer = _lib.KernelTimer()
		for i in range(steps2):
			step(args.warmup + args.steps + i)
		fence()
		if rank == 0:
			kt2 = _lib.timer.summary()
			_lib.timer = None
	if use_dist:
		t = torch.tensor([elapsed], dtype = torch.float64, device = device)
		dist.all_reduce(t, op = dist.ReduceOp.MAX)
		elapsed = float(t.item())

	if rank == 0:
		audio_s = world * BATCH * SECS * args.steps
		value = audio_s / elapsed
		roof = None
		main_name = main_family
		fused_name = main_name + '+bn_bwd'  # the same kernel symbol launched as a dgrad with the fused BN-backward epilogue (functional._dgrad)
		if kt:  # every other family comes from the second pass (per-step figures use its own step count)
			for name, v in kt2.items():
				if name not in (main_name, fused_name):
					kt[name] = dict(v, launches = v['launches'] * args.steps // steps2, total_ms = v['total_ms'] * args.steps / steps2, work = v['work'] * args.steps / steps2, bytes = v['bytes'] * args.steps / steps2)
		plain = kt.get(main_name)
		if main_name in kt and fused_name in kt:
			a, f = kt[main_name], kt.pop(fused_name)
			kt[main_name] = dict(launches = a['launches'] + f['launches'], total_ms = a['total_ms'] + f['total_ms'], avg_us = 1e3 * (a['total_ms'] + f['total_ms']) / (a['launches'] + f['launches']), work = a['work'] + f['work'], bytes = a['bytes'] + f['bytes'])
		if main_name in kt:
			peak = PEAK_BF16_DENSE / 1e12 if args.dtype in ('bf16', 'f16') else 157.3  # (dense fp16 MFMA peak = the bf16 one)
			tf = lambda k: k['work'] / (k['total_ms'] * 1e-3) / 1e12
			k = kt[main_name]
			roof = dict(bound = 'mfma', kernel = f'conv1d_igemm_v2s_kernel<O, false / true> ({0 if plain is None else plain["launches"] // args.steps} forward + {(k["launches"] - (0 if plain is None or plain is k else plain["launches"])) // args.steps} dgrad launches per step; the dgrads, instantiation <.., true>, also run pass 1 of the BN backward of the layer below in their epilogue; the prologue conv runs here as its stride-2 fold)' if args.dtype in ('bf16', 'f16') else 'conv1d_igemm_kernel<float> (every forward + dgrad launch; exact-fp32 v_mfma_f32_32x32x2_f32)', achieved = round(tf(k), 2), peak = peak, unit = 'TFLOP/s', frac = round(tf(k) / peak, 4), traffic = None, traffic_source = None, algorithmic_mb_per_launch = round(k['bytes'] / k['launches'] / 1e6, 1), launches_per_step = k['launches'] // args.steps, avg_launch_us = round(k['avg_us'], 2), ms_per_step = round(k['total_ms'] / args.steps, 3), timing = f'HIP events on the launching stream around every launch of this kernel inside the timed region ({args.steps} steps); wgrad / conv_stack / hbm_kernels: the same way in a second pass of {steps2} steps right after it')
			hbm = {name[4:]: v for name, v in kt.items() if name.startswith('hbm:')}
			kt = {name: v for name, v in kt.items() if not name.startswith('hbm:')}
			if plain is not None and plain is not k:
				roof['plain_launches'] = dict(note = 'the launches of the same kernel without the fused BN-backward epilogue (forward, and the dgrads whose consumer is not fused): the epilogue adds work that is not counted as FLOPs', achieved = round(tf(plain), 2), frac = round(tf(plain) / peak, 4), launches_per_step = plain['launches'] // args.steps, avg_launch_us = round(plain['avg_us'], 2))
			others = {name: v for name, v in kt.items() if name != main_name}
			if 'conv1d_wgrad' in others:
				w = others['conv1d_wgrad']
				roof['wgrad'] = dict(kernel = 'conv1d_wgrad_v2_kernel incl. its split-K combine (+ general wgrad kernel on 3 small layers)', achieved = round(tf(w), 2), frac = round(tf(w) / peak, 4), launches_per_step = w['launches'] // args.steps, avg_launch_us = round(w['avg_us'], 2), ms_per_step = round(w['total_ms'] / args.steps, 3))
			allc = [v for v in kt.values()]
			roof['conv_stack'] = dict(achieved = round(sum(v['work'] for v in allc) / (sum(v['total_ms'] for v in allc) * 1e-3) / 1e12, 2), ms_per_step = round(sum(v['total_ms'] for v in allc) / args.steps, 3))
			roof['conv_stack']['frac'] = round(roof['conv_stack']['achieved'] / peak, 4)
			roof['whole_step_frac'] = round(FLOP_PER_AUDIO_S_FWD_BWD * value / world / (peak * 1e12), 4)
			# the HBM-bound kernels of the path (frontend, BN + activation passes): algorithmic bytes / HIP-event time against 8 TB/s
			roof['hbm_kernels'] = {name: dict(achieved = round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9, 1), peak = PEAK_HBM_GBS, unit = 'GB/s', frac = round(v['bytes'] / (v['total_ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), launches_per_step = v['launches'] // args.steps, ms_per_step = round(v['total_ms'] / args.steps, 3)) for name, v in hbm.items()}
			if 'logmel_kernel' in roof['hbm_kernels']:
				roof['hbm_kernels']['logmel_kernel']['note'] = 'FFT-issue bound (three radix-8 Stockham passes through LDS per pair of frames), not HBM bound: under 1 % of the step'
		line = dict(metric = 'audio-seconds/sec/node (fwd+bwd+CTC) at bs64x15s', value = round(value, 1), unit = 'audio-seconds/sec', n_gpus = world, steps = args.steps, warmup = args.warmup, ms_per_step = round(1e3 * elapsed / args.steps, 3), higher_is_better = True, scaling = 'weak', vs_baseline = None, dtype = args.dtype, data = 'synthetic', config = dict(workload = f'Wav2Letter full (18 conv + decoder, 66.5M params), {BATCH}x{SECS}s 16kHz per GPU, logmel+convstack+CTC fwd+bwd+clip+SGD, dropout {args.dropout:g}', global_batch = BATCH * world, parallelism = f'dp{world}'), loss = round(float(last['loss']), 4), loss_scaler = scaler_info, dist = dist_info, roofline = roof, parity = None)
	if use_dist:
		dist.destroy_process_group()
	if rank == 0:
		# the legs below start child processes / use the host cores: model, optimizer state and workspaces are released first
		del model, flat, opt, engine, x, xlen, y, ylen, last
		torch.cuda.empty_cache()
		if world == 1 and roof is not None and args.dtype in ('bf16', 'f16') and not args.no_traffic:
			traffic, src = measure_traffic(args, sequence, args.steps)
			roof['traffic'], roof['traffic_source'] = traffic, src  # (None + the reason when the counters could not be collected: no stale fallback)
		if world == 1 and not args.no_cpu_baseline:
			ref = {}
			line['cpu_baseline'] = cpu_baseline(keep = ref)
			line['parity'] = gpu_parity(ref, device)
			line['parity']['headline_dtype'] = args.dtype
		import ctypes
		ctypes.CDLL(None).fflush(None)
		sys.stdout.flush()
		os.dup2(real_stdout, 1)
		print(json.dumps(line), flush = True)
	return 0


if __name__ == '__main__':
	sys.exit(main())
