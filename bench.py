"""bench.py -- BASELINE.json's headline metric on the MI355X-native path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload wav2letter|jasper_large]

N > 1 without WORLD_SIZE in the environment: this process touches no GPU and starts its own N ranks as a CHILD
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`, the role of train.py:1057-1073's mp.spawn), relays rank 0's
JSON line and exits with the child's return code (the child tree is killed and 124 returned when it outlives
CONVASR_LAUNCH_TIMEOUT seconds).  Under torch.distributed.run (WORLD_SIZE set) it is one rank of that job.

A step = one full training iteration of the hot path on one synthetic batch per GPU (train.py:745-783 of the reference):
logmel frontend -> instance norm -> conv stack -> 1x1 decoder -> log-softmax -> CTC loss -> backward (dgrad / wgrad / BN / CTC)
-> gradient all-reduce (N > 1) -> clip_grad_norm_ -> optimizer.

--workload wav2letter (the default, the HEADLINE): BASELINE configs[2] / [3] -- Wav2Letter full (18 x Conv1d+BN+hardtanh+dropout
+mask), 64 utterances x 15 s of 16 kHz audio per GPU, bf16 MFMA convolutions with fp32 accumulation and fp32 master weights,
dropout 0.2, SGD (--dtype f16: fp16 storage + MFMA under apex's dynamic loss scaling; --dtype f32: the exact-fp32 parity path).
--workload jasper_large (NOT the headline): BASELINE configs[4] -- JasperNetLarge ("Jasper 10x5", models.py:1407-1409, dense
residuals), 32 utterances of 5-20 s per batch from BucketingBatchSampler -> collate_gpu (mixed lengths, bucketed as
train.py:597-601), fp16 under apex O2 loss scaling, NovoGrad.

Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0; its `parity` object is the second half of
BASELINE's metric: the per-utterance CTC losses of every GPU compute type relative to the CPU oracle on BASELINE configs[2]'s own batch
(64 x 15 s, lengths linspace(0.5, 1): the maximum over the 64 utterances) and on the 4-utterance sample the cpu_baseline leg runs anyway,
plus, for a bf16 headline, short timed regions of the same workload in the other compute types on the same device: fp16 (apex O2;
2.0e-4 at 64 x 15 s, above north_star's 1e-4), bf16x3 (--dtype bf16x3: fp32 storage, split-operand convs on the 16-bit matrix pipe,
csrc/split3.hip -- the path that meets 1e-4 at MFMA rate) and exact fp32 (--dtype f32: the v_mfma_f32 parity path).
"""
import argparse
import datetime
import json
import os
import signal
import socket
import subprocess
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL / cross-process GPU buffers need it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE, SECS, BATCH = 16000, 15, 64
# (the Wav2Letter conv stack is 19.98 GFLOP per audio-second, 2 * MAC, fwd + dgrad + wgrad; conv_stack_flops() below
# derives the same figure -- and configs[4]'s 27.7 GFLOP per audio-second forward -- from the module tree, tests/test_parallel_cpu.py)
PEAK_BF16_DENSE = 2.5e15  # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak
PEAK_F32_MFMA = 157.3e12  # exact-fp32 MFMA
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s
# demangled or (where the profiler's demangler does not know _Float16) mangled; <H, H, *> only -- NOT <H, float, 0>, the decoder head
MAIN_KERNEL_SYMBOLS = dict(
	bf16 = ('conv1d_igemm_v2s_kernel<unsigned short, unsigned short', 'conv1d_igemm_v2s_kernelIttLi'),
	f16 = ('conv1d_igemm_v2s_kernel<_Float16, _Float16', 'conv1d_igemm_v2s_kernel<__half, __half', 'conv1d_igemm_v2s_kernelIDF16_DF16_Li'))
MAIN_FAMILY = 'conv1d_igemm_v2s_kernel<bf16>'  # (family labels are shared by the two 16-bit types)
SPLIT_FAMILY, SPLIT_WGRAD_FAMILY, SPLIT_DTYPES = 'conv1d_igemm_v2s_kernel<x3>', 'conv1d_wgrad<x3>', ('bf16x3', 'f16x3', 'bf16x3f', 'f16x3f')  # convasr_amd.functional: the split-operand launches are booked apart, with their ALGORITHMIC FLOPs (each runs three MFMAs per product)


def graph_policy(opt, workload, gpus):
	"""--graph auto: step graphs for jasper_large on one rank."""
	return opt == 'on' or (opt == 'auto' and workload == 'jasper_large' and gpus == 1 and os.environ.get('CONVASR_FORCE_DIST') != '1')


def parse_args(argv = None):
	ap = argparse.ArgumentParser()
	ap.add_argument('--gpus', type = int, default = 1)
	ap.add_argument('--steps', type = int, default = 10)
	ap.add_argument('--warmup', type = int, default = 3)
	ap.add_argument('--workload', default = 'wav2letter', choices = ['wav2letter', 'jasper_large'])
	ap.add_argument('--dtype', default = None, choices = ['bf16', 'f16', 'f32', 'bf16x3', 'f16x3', 'bf16x3f', 'f16x3f'], help = 'default: bf16 (wav2letter), f16 (jasper_large); bf16x3 / f16x3: fp32 storage with split-operand convs (three 16-bit MFMAs per product, fp32-class accuracy); bf16x3f / f16x3f: that forward (the same loss), one 16-bit product per gradient in the backward')
	ap.add_argument('--dropout', type = float, default = 0.2, help = 'the reference default (train.py:1033) is 0.2; other values are for experiments only')
	ap.add_argument('--batch', type = int, default = None, help = 'TEST ONLY: utterances per GPU (a line measured with it is not the headline)')
	ap.add_argument('--secs', type = int, default = None, help = 'TEST ONLY: seconds per utterance (wav2letter)')
	ap.add_argument('--no-cpu-baseline', action = 'store_true')
	ap.add_argument('--no-kernel-timer', action = 'store_true')
	ap.add_argument('--no-f16-leg', action = 'store_true', help = 'skip the second timed region in fp16 (parity.f16_value)')
	ap.add_argument('--no-parity-legs', action = 'store_true', help = 'skip the short timed regions in bf16x3 and exact fp32 (parity.bf16x3_value, parity.f32_value)')
	ap.add_argument('--no-traffic', action = 'store_true', help = 'skip the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic in this run')
	ap.add_argument('--side-stream', default = 'auto', choices = ['auto', 'on', 'off'], nargs = '?', const = 'on',
		help = 'run wgrad on a second HIP stream beside the dgrad of the same layer.  auto: on for jasper_large (launches of 0.5-3 rounds leave CUs idle: +2 %% measured), '
			'off for wav2letter (launches fill the chip: -1 %%).  With the side stream on, every per-kernel HIP-event duration (the dominant kernel included) is measured in the '
			'second pass, where the side stream is switched off again: overlapped launches would inflate each other')
	ap.add_argument('--graph', default = 'auto', choices = ['auto', 'on', 'off'], nargs = '?', const = 'on',
		help = 'replay the training step from HIP graphs, one per batch shape (convasr_amd.train.GraphedTrainStep); jasper_large batches are then padded to their bucket\'s '
			'ceiling (one shape per bucket).  auto: on for jasper_large with one rank (~530 launches and ~19 ms of Python per ~38 ms step; a replay costs the host ~0.4 ms, and the replayed step itself is ~2 %% slower than the eager one '
			'with the wgrad side stream, whose time the line also reports), off for wav2letter (~110 launches, GPU-bound either way: +-0 measured, and the dominant kernel stays event-timed '
			'inside the timed region) and for N > 1 (RCCL has not run under capture on this pool).  With graphs on, per-kernel HIP-event durations come from the second, eager pass')
	ap.add_argument('--no-jasper-leg', action = 'store_true', help = 'skip the bounded BASELINE configs[4] leg (extra.jasper_large) of the default line')
	ap.add_argument('--launcher-dry-run', action = 'store_true', help = 'test hook: ranks only rendezvous (gloo, CPU tensors) and rank 0 prints a line; exercises the self-launch path without a GPU')
	args = ap.parse_args(argv)
	if args.dtype is None:
		args.dtype = 'f16' if args.workload == 'jasper_large' else 'bf16'
	args.side_stream = args.side_stream == 'on' or (args.side_stream == 'auto' and args.workload == 'jasper_large')
	args.graph_opt = args.graph
	args.graph = graph_policy(args.graph_opt, args.workload, args.gpus)
	return args


# ------------------------------------------------------------------------------------------------ launcher (parent side)

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
	device on this image); the ranks run in a child process tree started by torch.distributed.run, in a session of its own so
	that a hung rendezvous / collective can be ended as a whole (a fresh child or an exit -- never a re-exec)."""
	if not args.launcher_dry_run and os.environ.get('CONVASR_SHARE_GPU') != '1':
		import torch
		have = torch.cuda.device_count()
		if have < args.gpus:
			print(f'bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible', file = sys.stderr)
			return 2
	cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
		'--master-port', str(_free_port()), os.path.abspath(__file__), *argv]
	limit = float(os.environ.get('CONVASR_LAUNCH_TIMEOUT', 1500))
	proc = subprocess.Popen(cmd, stdout = subprocess.PIPE, text = True, start_new_session = True)  # stderr is inherited
	try:
		out, _ = proc.communicate(timeout = limit)
	except subprocess.TimeoutExpired:
		print(f'bench.py: the {args.gpus} ranks did not finish within {limit:.0f} s (CONVASR_LAUNCH_TIMEOUT): ending the child tree', file = sys.stderr)
		for sig in (signal.SIGTERM, signal.SIGKILL):
			try:
				os.killpg(proc.pid, sig)  # start_new_session: the child leads its own process group
			except ProcessLookupError:
				break
			try:
				proc.wait(timeout = 10)
				break
			except subprocess.TimeoutExpired:
				continue
		return 124
	line = _last_json_line(out or '')
	if line is not None:
		print(line, flush = True)
	elif out:
		sys.stderr.write(out)
	if proc.returncode != 0:
		return proc.returncode
	return 0 if line is not None else 1


# ------------------------------------------------------------------------------------------------ rank pre-flight (before any GPU call)

def _kfd_gpu_nodes():
	"""DRM render minors of the GPUs in the order the HIP runtime enumerates them: the GPU nodes of the KFD topology in node order; where an
	unprivileged process may not read the topology (the pool's containers), the AMD render nodes in PCI-address order, which is the same
	order on a single-root node."""
	base = '/sys/class/kfd/kfd/topology/nodes'
	try:
		out = []
		for n in sorted((d for d in os.listdir(base) if d.isdigit()), key = int):
			props = dict(ln.split(None, 1) for ln in open(os.path.join(base, n, 'properties')).read().splitlines() if ' ' in ln)
			if int(props.get('simd_count', '0')) > 0:
				out.append(int(props.get('drm_render_minor', '-1')))
		if out:
			return out
	except OSError:
		pass
	import glob
	nodes = []
	for r in glob.glob('/sys/class/drm/renderD*'):
		try:
			if open(os.path.join(r, 'device', 'vendor')).read().strip() == '0x1002':
				nodes.append((os.path.basename(os.path.realpath(os.path.join(r, 'device'))), int(os.path.basename(r)[7:])))
		except OSError:
			continue
	if not nodes:
		raise FileNotFoundError('no KFD topology and no AMD render node under /sys/class/drm')
	return [minor for _, minor in sorted(nodes)]


def _cpulist(text):
	cpus = set()
	for part in text.strip().split(','):
		if part:
			lo, _, hi = part.partition('-')
			cpus.update(range(int(lo), int(hi or lo) + 1))
	return cpus


def pin_to_gpu_numa_node(local_rank):
	"""os.sched_setaffinity to the cores local to this rank's GPU (the NUMA node of its PCIe root, /sys/class/drm/renderD*/device/
	{numa_node, local_cpulist}) -- host launch latency and the pinned staging buffers then stay on the socket the GPU hangs off.
	Pure sysfs reads: runs before the first GPU call.  Best effort: returns a dict saying what was done (or why not)."""
	step = 'reading the KFD topology'
	try:
		nodes = _kfd_gpu_nodes()
		vis = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('ROCR_VISIBLE_DEVICES')
		if vis and all(v.strip().isdigit() for v in vis.split(',')):
			nodes = [nodes[int(v)] for v in vis.split(',') if int(v) < len(nodes)]
		import torch
		visible = torch.cuda.device_count()  # (does not initialise the GPU on this image)
		if visible != len(nodes):  # a container that is handed some of the host's GPUs still sees all of them in sysfs: which one is device i?
			return dict(pinned = False, reason = f'{len(nodes)} GPU render nodes in sysfs for {visible} visible device(s): the mapping is ambiguous')
		minor = nodes[local_rank]
		step = 'reading the render node\'s numa_node / local_cpulist'
		dev = f'/sys/class/drm/renderD{minor}/device'
		node = int(open(os.path.join(dev, 'numa_node')).read())
		cpus = _cpulist(open(os.path.join(dev, 'local_cpulist')).read()) & os.sched_getaffinity(0)
		if not cpus:
			return dict(pinned = False, reason = 'no local cpulist', numa_node = node)
		step = f'sched_setaffinity to {len(cpus)} cpus of node {node}'
		os.sched_setaffinity(0, cpus)
		return dict(pinned = True, numa_node = node, cpus = len(cpus), render_minor = minor)
	except Exception as e:  # no sysfs topology (container), odd masks, a sandbox that forbids the call: the run goes on unpinned
		return dict(pinned = False, reason = f'{step}: {type(e).__name__}: {e}')


# ------------------------------------------------------------------------------------------------ workloads

def synthetic_batch(device, batch = BATCH, secs = SECS, seed = 1):
	import torch
	g = torch.Generator().manual_seed(seed)
	x = torch.rand(batch, SAMPLE_RATE * secs, generator = g) * 2 - 1
	xlen = torch.ones(batch)
	y = torch.randint(0, 37, (batch, 1, 10 * secs), generator = g)
	ylen = torch.full((batch, 1), 10 * secs, dtype = torch.long)
	return tuple(t.to(device) for t in (x, xlen, y, ylen))


def conv_stack_flops(model, batch, samples):
	"""Algorithmic FLOPs (2 x MAC) of one training step's convolutions on a (batch, samples) waveform batch: forward + weight gradient
	for every conv (residual 1x1 convs and the decoder included), + input gradient for all but the prologue."""
	import torch.nn as nn
	t = 1 + samples // model.frontend.hop_length  # frames
	first = model.backbone[0].conv[0][-1]
	fwd = bwd = 0.0

	def book(c, tin):
		nonlocal fwd, bwd
		tout = (tin + 2 * c.padding[0] - c.dilation[0] * (c.kernel_size[0] - 1) - 1) // c.stride[0] + 1
		f = 2.0 * batch * tout * c.out_channels * (c.in_channels // c.groups) * c.kernel_size[0]
		fwd += f
		bwd += f if c is first else 2 * f
		return tout
	for blk in model.backbone:
		for seq in blk.conv:
			for c in seq:
				if isinstance(c, nn.Conv1d):
					t = book(c, t)
		for c in blk.conv_residual:  # 1x1 convs on earlier block outputs: same frame count (every block but the prologue keeps it)
			if isinstance(c, nn.Conv1d):
				book(c, t)
	for c in model.decoder.modules():
		if isinstance(c, nn.Conv1d):
			book(c, t)
	return fwd, bwd


class Workload:
	"""What one rank steps through: model, optimizer, the device-resident batches, and the audio / FLOPs each of them carries."""

	def __init__(self, args, device, rank, world, dtype = None):
		import torch
		import convasr_amd as ca
		self.args, self.device = args, device
		dtype = dtype or args.dtype
		self.dtype = dtype
		compute = dict(bf16 = torch.bfloat16, f16 = torch.float16, f32 = torch.float32, bf16x3 = 'bf16x3', f16x3 = 'f16x3', bf16x3f = 'bf16x3f', f16x3f = 'f16x3f')[dtype]
		fe = ca.models.LogFilterBankFrontend(64, SAMPLE_RATE, 0.02, 0.01, 'hann_window')
		n_batches = args.warmup + args.steps + min(args.steps, 5)
		if args.workload == 'wav2letter':
			self.batch, self.secs = args.batch or BATCH, args.secs or SECS
			model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = args.dropout, check_time_dim_padded = False, compute_dtype = compute)
			self.model = model.to(device).train()
			self.flat = ca.train.FlatParameters(self.model)
			self.opt = ca.train.SGD(self.flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)  # train.py:879-884 defaults
			b = synthetic_batch(device, batch = self.batch, secs = self.secs, seed = 1 + rank)
			self.batches = [b]
			self.audio = [(self.batch * self.secs, self.batch * self.secs)]  # (unpadded, padded) audio seconds per batch
			self.name = f'Wav2Letter full (18 conv + decoder, 66.5M params), {self.batch}x{self.secs}s 16kHz per GPU, logmel+convstack+CTC fwd+bwd+clip+SGD, dropout {args.dropout:g}'
		else:
			self.batch = args.batch or 32
			model = ca.models.JasperNetLarge(64, [38], frontend = fe, dropout = args.dropout, check_time_dim_padded = False, compute_dtype = compute)
			self.model = model.to(device).train()
			self.flat = ca.train.FlatParameters(self.model)
			self.opt = ca.optimizers.NovoGrad(self.flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
			# mixed lengths, bucketed: every rank draws its own batches of one schedule (DistributedSamplerWrapper: the W ranks of an
			# iteration get batches of the same bucket, i.e. equal padded length -- balanced steps)
			ds = ca.datasets.SyntheticAudioTextDataset(self.batch * world * n_batches * 2, min_duration = 5.0, max_duration = 20.0, seed = 7)
			sampler = ca.datasets.BucketingBatchSampler(ds, batch_size = self.batch, world_size = world)
			sampler.set_epoch(0)
			if world > 1:
				sampler = ca.datasets.DistributedSamplerWrapper(sampler, num_replicas = world, rank = rank)
			self.batches, self.audio = [], []
			for meta, s, x, xlen, y, ylen in ca.datasets.gpu_batches(ds, sampler, device, pad_to_bucket = bool(args.graph)):
				self.batches.append((x, xlen, y, ylen))
				self.audio.append((sum(m['duration'] for m in meta), x.shape[0] * x.shape[1] / SAMPLE_RATE))
				if len(self.batches) == n_batches:
					break
			self.name = f'JasperNetLarge (Jasper 10x5, dense residuals, {sum(p.numel() for p in model.parameters()) / 1e6:.0f}M params), {self.batch} utterances of 5-20 s per GPU and step (BucketingBatchSampler -> collate_gpu, mixed lengths), logmel+convstack+CTC fwd+bwd+clip+NovoGrad, dropout {args.dropout:g}'
		self.model._convasr_flat = self.flat
		if dtype in ('f16x3', 'f16x3f'):  # fp16 planes: the output gradients need the dynamic loss scaler like plain fp16's
			ca.models.data_parallel_and_autocast(self.model, self.opt, compute_dtype = dtype)
			assert self.model.split_dtype == torch.float16 and self.flat.loss_scaler is not None
		if dtype == 'f16':  # apex O2: fp16 compute, fp32 masters, dynamic loss scaling from 2^16 (the start-up overflows fall into the warm-up steps)
			ca.models.data_parallel_and_autocast(self.model, self.opt, opt_level = 'O2')
			assert self.model.compute_dtype == torch.float16 and self.flat.loss_scaler is not None
		self.flops = [conv_stack_flops(self.model, b[0].shape[0], b[0].shape[1]) for b in self.batches]
		self.stepper = None

	def make_stepper(self, engine, world):
		"""What a step is: train_step on the engine (N > 1, or --graph off), or the same step replayed from one HIP graph per batch shape."""
		import convasr_amd as ca
		graphed = bool(self.args.graph) and (engine is self.model or engine.capturable)  # (a data-parallel engine is captured with its RCCL collectives)
		self.stepper = ca.train.GraphedTrainStep(engine, self.opt, max_norm = 100.0, warmup = 1, enabled = graphed, world_size = world, sync_metrics = engine is not self.model)

		def step(i):
			x, xlen, y, ylen = self.batches[self.batch_of(i)]
			if self.stepper.enabled:
				return self.stepper(x, xlen, y, ylen, iteration = i)
			return ca.train.train_step(engine, self.opt, x, xlen, y, ylen, world_size = world, iteration = i, sync_metrics = engine is not self.model)
		return step

	def prime_graphs(self, step, indices):
		"""Extra untimed warm-up: every batch shape the steps `indices` will use gets its eager warm-up step(s) and its capture now."""
		if self.stepper is None or not self.stepper.enabled:
			return 0
		extra, seen = 0, set()
		for i in indices:
			key = self.stepper.key_of(*self.batches[self.batch_of(i)])
			if key in seen:
				continue
			seen.add(key)
			while self.stepper.inputs(key) is None:
				step(i)
				extra += 1
		return extra

	def batch_of(self, i):
		return i % len(self.batches)

	def release(self):
		self.model = self.flat = self.opt = self.batches = self.stepper = None


def run_timed(args, wl, engine, world, fence, time_main_kernel, on_warm = None, probe = None):
	"""W warm-up steps, then K timed steps between two fences.  Returns (elapsed s, audio (unpadded, padded), flops (fwd, bwd), last
	result, kernel-timer summary, launch sequence)."""
	import convasr_amd as ca
	from convasr_amd import _lib
	last = None
	step = wl.make_stepper(engine, world)
	for i in range(args.warmup):
		last = step(i)
	run_timed.extra_warmup = wl.prime_graphs(step, range(args.warmup, args.warmup + args.steps + min(args.steps, 5)))
	fence()
	run_timed.warm_value = on_warm() if on_warm is not None else None  # (a host read between warm-up and timed region, e.g. the loss scaler's overflow count)
	# HIP events (on the launching stream) bracket every launch of the DOMINANT kernel inside the timed region.  The other kernel
	# families (wgrad, the HBM-bound passes, the small layers) are event-timed in a second, untimed pass of a few steps right after
	# it: an event pair costs ~5 us of stream time, and bracketing all ~110 launches of a step slowed the headline by 3.4 %
	# (18.06 vs 17.47 ms per step on one device; bracketing the dominant kernel only: ~1 %).
	if time_main_kernel:
		_lib.timer = _lib.KernelTimer(only = [SPLIT_FAMILY] if wl.dtype in SPLIT_DTYPES else [MAIN_FAMILY, MAIN_FAMILY + '+bn_bwd'] if wl.dtype != 'f32' else ['conv1d_igemm (other variants)'])
	if hasattr(engine, 'exposed_comm_events'):
		engine.exposed_comm_events = []
	if probe is not None:
		probe.start()
	t0 = time.perf_counter()
	for i in range(args.steps):
		last = step(args.warmup + i)
	fence()
	elapsed = time.perf_counter() - t0
	if probe is not None:
		probe.stop()
	kt = _lib.timer.summary() if _lib.timer is not None else {}
	sequence = list(_lib.timer.sequence) if _lib.timer is not None else []
	_lib.timer = None
	idx = [wl.batch_of(args.warmup + i) for i in range(args.steps)]
	audio = tuple(sum(wl.audio[j][k] for j in idx) for k in (0, 1))
	flops = tuple(sum(wl.flops[j][k] for j in idx) for k in (0, 1))
	return elapsed, audio, flops, last, kt, sequence, step


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
	return dict(value = round(batch * secs / mean, 2), best = round(batch * secs / best, 2), unit = 'audio-seconds/sec', cores = torch.get_num_threads(),
		host_cpus = os.cpu_count(), kind = 'port',
		sample = f'{batch}x{secs}s utterances, Wav2Letter full fp32, fwd+CTC+bwd+clip+SGD, mean of {iters} timed iterations after 1 warm-up ({mean:.2f} s/step mean, {best:.2f} s/step best)')


PARITY_DTYPES = ('f32', 'bf16x3', 'f16x3', 'f16', 'bf16')


def parity_reference(batch = BATCH, secs = SECS):
	"""The CPU oracle's per-utterance CTC losses on BASELINE configs[2]'s own batch -- 64 x 15 s, lengths linspace(0.5, 1), ~8 labels per
	valid second (the batch of tests/test_full_size_and_step_graphs_gpu.py's full-size test) -- from one train-mode forward pass (models.py:282-326; ~5 s on 16
	host threads).  Test infrastructure used as the checker: nothing here is timed or shipped."""
	import torch
	from oracle import convasr_oracle as O
	g = torch.Generator().manual_seed(11)
	x = torch.rand(batch, SAMPLE_RATE * secs, generator = g) * 2 - 1
	xlen = torch.linspace(0.5, 1, batch)
	y = torch.randint(0, 37, (batch, 1, 10 * secs), generator = g)
	ylen = (xlen * 8 * secs).long().clamp(min = 1).view(batch, 1)
	plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
	fe = O.frontend_config()
	sd = O.init_state_dict(plan, seed = 1, frontend = fe)
	t0 = time.perf_counter()
	with torch.no_grad():
		out = O.jasper_forward({k: v.clone() for k, v in sd.items()}, plan, x, xlen, y, ylen, frontend = fe, training = True)
	return dict(sd = sd, batch = (x, xlen, y, ylen), loss_vec = out['loss'].float().clone(), oracle_forward_s = round(time.perf_counter() - t0, 2))


def _gpu_losses(ref, device, dtypes):
	import torch
	import convasr_amd as ca
	x, xlen, y, ylen = (t.to(device) for t in ref['batch'])
	out = {}
	for name in dtypes:
		dt = dict(f32 = torch.float32, bf16 = torch.bfloat16, f16 = torch.float16).get(name, name)
		fe = ca.models.LogFilterBankFrontend(64, SAMPLE_RATE, 0.02, 0.01, 'hann_window')
		model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.0, check_time_dim_padded = False, compute_dtype = dt)
		missing = model.load_state_dict(ref['sd'], strict = False)
		assert not missing.missing_keys, missing
		model.to(device).train()
		# (the split-operand convs are the TRAINING path's: they run where a gradient is wanted, so their forward is taken with autograd on)
		with (torch.enable_grad() if name in SPLIT_DTYPES else torch.no_grad()):
			loss = model(x, xlen, y = y, ylen = ylen)['loss'].detach().float().cpu()
		out[name] = float(((loss - ref['loss_vec']).abs() / ref['loss_vec'].abs()).max())
		del model, loss
		torch.cuda.empty_cache()
	return {k: float(f'{v:.3e}') for k, v in out.items()}


def gpu_parity(ref, device, ref_full = None):
	"""BASELINE's "CTC loss rel-err vs ref": the per-utterance CTC losses of the MI355X path, for every compute type, against the CPU
	oracle's (same inputs, same initial parameters, train-mode batch statistics, dropout 0 -- the oracle has no dropout; the headline
	throughput runs with 0.2).  Relative error = max over utterances.  ctc_loss_rel_err: BASELINE configs[2]'s own 64 x 15 s batch with
	lengths linspace(0.5, 1) (ref_full, parity_reference()); ctc_loss_rel_err_sample: the 4 full-length utterances of the cpu_baseline leg."""
	out = dict(north_star_bound = 1e-4, reference = 'oracle (fp32 CPU restatement of the reference path, pinned to the reference by tests/golden)')
	x = ref['batch'][0]
	small = _gpu_losses(ref, device, PARITY_DTYPES)
	if ref_full is not None:
		xf = ref_full['batch'][0]
		out['ctc_loss_rel_err'] = _gpu_losses(ref_full, device, PARITY_DTYPES)
		out['sample'] = f'{xf.shape[0]}x{xf.shape[1] // SAMPLE_RATE}s utterances, lengths linspace(0.5, 1), ~8 labels per valid second: BASELINE configs[2]\'s own batch size; same initial parameters, train-mode forward, dropout 0; per-utterance CTC loss, MAXIMUM relative error over the {xf.shape[0]} utterances (oracle forward: {ref_full["oracle_forward_s"]} s of CPU)'
		out['ctc_loss_rel_err_sample'] = small
		out['sample_small'] = f'{x.shape[0]}x{x.shape[1] // SAMPLE_RATE}s full-length utterances of the cpu_baseline leg (the easy case: no masked tail)'
	else:
		out['ctc_loss_rel_err'] = small
		out['sample'] = f'{x.shape[0]}x{x.shape[1] // SAMPLE_RATE}s utterances of the cpu_baseline leg, same initial parameters, train-mode forward, dropout 0; per-utterance CTC loss, max relative error'
	best = out['ctc_loss_rel_err']
	out['within_bound'] = [k for k in PARITY_DTYPES if best[k] <= 1e-4]
	out['note'] = ('f32 (exact-fp32 MFMA) and the split-operand types bf16x3 / f16x3 (fp32 storage, three 16-bit MFMAs per product: csrc/split3.hip) are within north_star\'s 1e-4; '
		'bf16 (8 significant bits of storage) and f16 (11) buy their throughput at the error shown: the deviation is the storage type\'s own '
		'(tests/test_bf16_parity_gpu.py: a CPU restatement with the same storage type deviates alike)')
	return out


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
			cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', d, '--', sys.executable, os.path.abspath(__file__),
				'--steps', '1', '--warmup', '1', '--workload', args.workload, '--dtype', args.dtype, '--dropout', str(args.dropout),
				'--no-cpu-baseline', '--no-kernel-timer', '--no-traffic', '--no-f16-leg', '--no-parity-legs', '--no-jasper-leg', '--graph', 'off', '--side-stream', 'off']
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
	src = (f'measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child passes over bench.py --steps 1 --warmup 1, '
		f'mean of {out["FETCH_SIZE"][1]} dispatches = {out["FETCH_SIZE"][2]}, MB per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024')
	return round(mb, 1), src


def dry_run_rank(args):
	"""--launcher-dry-run: rendezvous + one all-reduce on CPU tensors (gloo); rank 0 prints the line the parent relays."""
	import torch
	import torch.distributed as dist
	dist.init_process_group('gloo', timeout = datetime.timedelta(seconds = 120))
	rank, world = dist.get_rank(), dist.get_world_size()
	if int(os.environ.get('CONVASR_DRY_RUN_HANG_RANK', '-1')) == rank:
		time.sleep(3600)  # test hook: the parent's CONVASR_LAUNCH_TIMEOUT must end the tree
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
		print(json.dumps(dict(metric = 'launcher-dry-run', value = float(t), unit = 'sum of rank+1', n_gpus = world, steps = args.steps, warmup = args.warmup,
			dist = dict(backend = backend, world_size = world, launcher = os.environ.get('TORCHELASTIC_RUN_ID') is not None))), flush = True)
	return 0


def roofline_of(args, wl, kt, kt2, steps2, value, world, graphed = False):
	"""The `roofline` object: the dominant kernel from the HIP events of the timed region (kt), every other family from the second
	pass (kt2, steps2 steps), all from algorithmic FLOPs / bytes booked per launch by convasr_amd.ops."""
	main_name, fused_name = MAIN_FAMILY, MAIN_FAMILY + '+bn_bwd'
	split = wl.dtype in SPLIT_DTYPES
	half = wl.dtype in ('bf16', 'f16') or split  # (priced against the 16-bit dense MFMA peak)
	if split:
		main_name = SPLIT_FAMILY
	elif not half:
		main_name = 'conv1d_igemm (other variants)'
	scale = args.steps / max(steps2, 1)
	second_pass_main = main_name not in kt and main_name in kt2  # (side stream on: nothing was event-timed inside the timed region)
	if second_pass_main:
		kt = dict(kt)
		for name in (main_name, fused_name):
			if name in kt2:
				v = kt2[name]
				kt[name] = dict(v, launches = int(v['launches'] * scale), total_ms = v['total_ms'] * scale, work = v['work'] * scale, bytes = v['bytes'] * scale)
	if main_name not in kt:
		return None
	for name, v in kt2.items():  # per-step figures of the second pass, rescaled to the timed region's step count
		if name not in (main_name, fused_name):
			kt[name] = dict(v, launches = int(v['launches'] * scale), total_ms = v['total_ms'] * scale, work = v['work'] * scale, bytes = v['bytes'] * scale)
	plain = kt.get(main_name)
	if fused_name in kt:
		a, f = kt[main_name], kt.pop(fused_name)
		n, ms = a['launches'] + f['launches'], a['total_ms'] + f['total_ms']
		kt[main_name] = dict(launches = n, total_ms = ms, avg_us = 1e3 * ms / n, work = a['work'] + f['work'], bytes = a['bytes'] + f['bytes'])
	peak = (PEAK_BF16_DENSE if half else PEAK_F32_MFMA) / 1e12  # (dense fp16 MFMA peak = the bf16 one)
	tf = lambda k: k['work'] / (k['total_ms'] * 1e-3) / 1e12
	k = kt[main_name]
	n_plain = 0 if plain is None or plain is k else plain['launches']
	hi_bwd = wl.dtype.endswith('x3f')
	if split:
		kernel = (f'conv1d_igemm_v2s_kernel<H, float, 0> as a split-operand conv ({k["launches"] // args.steps} ' + ('forward launches per step (the dgrads run as one 16-bit product: roofline.dgrad_one_product)' if hi_bwd else 'forward + dgrad launches per step') + ': 16-bit planes (hi, lo, hi) of the fp32 operands read as 3 C channels, fp32 output; '
			'achieved / frac count ALGORITHMIC FLOPs, every product costs three MFMAs: executed_mfma_frac = 3 x frac; the strided prologue runs here as its stride-1 fold, the 38-class head as a 128-class problem)')
	elif half:
		kernel = (f'conv1d_igemm_v2s_kernel<O, BNF> ({(plain["launches"] if plain else 0) // args.steps} forward / plain + {(k["launches"] - n_plain) // args.steps} fused dgrad launches per step; '
			'the fused dgrads also run pass 1 of the BN backward of the layer below in their epilogue; the prologue conv runs here as its stride-2 fold)')
	else:
		kernel = 'conv1d_igemm_kernel<float> (every forward + dgrad launch; exact-fp32 v_mfma_f32_32x32x2_f32)'
	roof = dict(bound = 'mfma', kernel = kernel, achieved = round(tf(k), 2), peak = peak, unit = 'TFLOP/s', frac = round(tf(k) / peak, 4), traffic = None, traffic_source = None,
		algorithmic_mb_per_launch = round(k['bytes'] / k['launches'] / 1e6, 1), launches_per_step = k['launches'] // args.steps, avg_launch_us = round(k['avg_us'], 2),
		ms_per_step = round(k['total_ms'] / args.steps, 3),
		timing = (f'HIP events on the launching stream around every launch, in a second pass of {steps2} steps right after the timed region (the timed region ' + ('replays the step from HIP graphs' if graphed else 'runs wgrad on a side stream') + ': no launch of it can be bracketed, and overlapped launches would inflate each other\'s durations; the second pass launches the same kernels eagerly on one stream)'
			if second_pass_main else f'HIP events on the launching stream around every launch of this kernel inside the timed region ({args.steps} steps); wgrad / conv_stack / hbm_kernels: the same way in a second pass of {steps2} steps right after it'))
	hbm = {name[4:]: v for name, v in kt.items() if name.startswith('hbm:')}
	kt = {name: v for name, v in kt.items() if not name.startswith('hbm:')}
	if plain is not None and plain is not k:
		roof['plain_launches'] = dict(note = 'the launches of the same kernel without the fused BN-backward epilogue (forward, and the dgrads whose consumer is not fused): the epilogue adds work that is not counted as FLOPs',
			achieved = round(tf(plain), 2), frac = round(tf(plain) / peak, 4), launches_per_step = plain['launches'] // args.steps, avg_launch_us = round(plain['avg_us'], 2))
	if split:
		roof['executed_mfma_frac'] = round(3 * tf(k) / peak, 4)
	wname = SPLIT_WGRAD_FAMILY if (split and not hi_bwd) else 'conv1d_wgrad'
	if hi_bwd and MAIN_FAMILY in kt:
		g = kt[MAIN_FAMILY]
		roof['dgrad_one_product'] = dict(kernel = 'conv1d_igemm_v2s_kernel<H, float, 0>: dy rounded once to 16 bits x w_hi, fp32 dx', achieved = round(tf(g), 2), frac = round(tf(g) / peak, 4), launches_per_step = g['launches'] // args.steps, avg_launch_us = round(g['avg_us'], 2), ms_per_step = round(g['total_ms'] / args.steps, 3))
	if wname in kt and kt[wname] is not k:
		w = kt[wname]
		roof['wgrad'] = dict(kernel = 'conv1d_wgrad_v2_kernel incl. its split-K combine' + (' over the hi plane of the saved planes, read in place (frames 3 Cin elements apart), x the 16-bit dy: one product' if hi_bwd else ' over the planes read as 3 T frames (three MFMAs per product: executed = 3 x achieved)' if split else ' (+ general wgrad kernel on small layers)'), achieved = round(tf(w), 2), frac = round(tf(w) / peak, 4),
			launches_per_step = w['launches'] // args.steps, avg_launch_us = round(w['avg_us'], 2), ms_per_step = round(w['total_ms'] / args.steps, 3))
	allc = list(kt.values())
	stack = sum(v['work'] for v in allc) / (sum(v['total_ms'] for v in allc) * 1e-3) / 1e12
	roof['conv_stack'] = dict(achieved = round(stack, 2), ms_per_step = round(sum(v['total_ms'] for v in allc) / args.steps, 3), frac = round(stack / peak, 4))
	if split:  # the MFMA work the stack EXECUTES: three products per algorithmic FLOP in the split-operand launches, one in the others ('...x3f': the backward)
		executed = sum(v['work'] * (3.0 if name in (SPLIT_FAMILY, SPLIT_WGRAD_FAMILY) else 1.0) for name, v in kt.items())
		roof['conv_stack']['executed_mfma_frac'] = round(executed / (sum(v['total_ms'] for v in allc) * 1e-3) / 1e12 / peak, 4)
	roof['whole_step_frac'] = round(value / world / (peak * 1e12), 4)  # (value here: algorithmic conv FLOP/s of the whole job)
	# the HBM-bound kernels of the path (frontend, BN + activation passes): algorithmic bytes / HIP-event time against 8 TB/s
	gbs = lambda v: v['bytes'] / (v['total_ms'] * 1e-3) / 1e9
	roof['hbm_kernels'] = {name: dict(achieved = round(gbs(v), 1), peak = PEAK_HBM_GBS, unit = 'GB/s', frac = round(gbs(v) / PEAK_HBM_GBS, 4),
		launches_per_step = v['launches'] // args.steps, ms_per_step = round(v['total_ms'] / args.steps, 3)) for name, v in hbm.items()}
	if 'logmel_kernel' in roof['hbm_kernels']:
		roof['hbm_kernels']['logmel_kernel']['note'] = 'FFT-issue bound (three radix-8 Stockham passes through LDS per pair of frames), not HBM bound: under 1 % of the step'
	return roof


# ------------------------------------------------------------------------------------------------ device state over the timed region

class DeviceProbe:
	"""Host-side sampler of THIS rank's card over the timed region, so that a round-to-round difference of the headline can be
	attributed from the line alone (the pool's devices differ by +-4 % on the MFMA kernels; the step is package-power limited):
	* hwmon (sysfs) every ~20 ms from a NATIVE thread (libconvasr_smi.so: a Python sampler thread took the GIL from the launching thread often
	  enough to cost the eager step ~1 %): power1_input -> power_w_mean, freq1_input -> sclk_mhz_mean;
	* the firmware's throttle accumulators from the gpu_metrics table (librocm_smi64 through convasr_amd/libconvasr_smi.so, ctypes) at
	  start() and stop(): ppt_residency = d ppt_residency_acc / d accumulation_counter, likewise thermal / PROCHOT.
	No HIP call is made here (the card is found by the PCI address torch already knows).  Every field is None when its source is not
	readable on this box; `sources` says which were."""

	def __init__(self, device):
		import ctypes
		import glob
		import torch
		self.hw, self.smi, self.dv, self.samples, self.acc = None, None, -1, [], [None, None]
		self.native, self.hw_means = False, None
		self.why = []
		try:
			props = torch.cuda.get_device_properties(device)
			dom, bus, dev = int(getattr(props, 'pci_domain_id', 0)), int(props.pci_bus_id), int(props.pci_device_id)
			prefix = '%04x:%02x:%02x' % (dom, bus, dev)
			cards = [c for c in glob.glob('/sys/class/drm/card*') if os.path.basename(os.path.realpath(c + '/device')).startswith(prefix)]
			hw = glob.glob(cards[0] + '/device/hwmon/hwmon*') if cards else []
			self.hw = hw[0] if hw else None
			if self.hw is None:
				self.why.append(f'no hwmon directory for PCI {prefix}')
			lib = os.path.join(ROOT, 'convasr_amd', 'libconvasr_smi.so')
			if os.path.exists(lib):
				self.smi = ctypes.CDLL(lib)
				self.smi.convasr_smi_open.restype, self.smi.convasr_smi_open.argtypes = ctypes.c_int, [ctypes.c_int] * 3
				self.smi.convasr_smi_sample.restype, self.smi.convasr_smi_sample.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
				self.smi.convasr_hwmon_start.restype, self.smi.convasr_hwmon_start.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]
				self.smi.convasr_hwmon_stop.restype, self.smi.convasr_hwmon_stop.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_double)]
				self.dv = self.smi.convasr_smi_open(dom, bus, dev)
				if self.dv < 0:
					self.why.append(f'rocm_smi: no device for PCI {prefix} (code {self.dv})')
			else:
				self.why.append('convasr_amd/libconvasr_smi.so not built')
		except Exception as e:
			self.why.append(f'{type(e).__name__}: {e}')

	def _read(self, name):
		try:
			return int(open(os.path.join(self.hw, name)).read())
		except Exception:
			return None

	def _metrics(self):
		import ctypes
		if self.smi is None or self.dv < 0:
			return None
		out = (ctypes.c_double * 10)()
		return list(out) if self.smi.convasr_smi_sample(self.dv, out) == 0 else None

	def start(self):
		self.samples, self.hw_means = [], None
		self.acc = [self._metrics(), None]
		self.native = False
		if self.hw is not None and self.smi is not None and hasattr(self.smi, 'convasr_hwmon_start'):
			self.native = self.smi.convasr_hwmon_start(self.hw.encode(), 20000) == 0  # a native thread: no Python (no GIL traffic) in the sampling loop; every 20 ms (every 5 ms cost the Wav2Letter step 0.4 %: each hwmon read is an SMU query, profiles/r05_ab_probe.txt)

	def stop(self):
		import ctypes
		self.acc[1] = self._metrics()
		if self.native:
			out = (ctypes.c_double * 3)()
			if self.smi.convasr_hwmon_stop(out) == 0:
				self.hw_means = (out[0] or None, out[1] or None, int(out[2]))
			self.native = False

	def summary(self):
		power, sclk, n_samples = self.hw_means if getattr(self, 'hw_means', None) else (None, None, 0)
		a, b = self.acc
		res = None
		if a is not None and b is not None and b[0] > a[0]:
			ticks = b[0] - a[0]
			res = dict(ppt = round((b[1] - a[1]) / ticks, 4), prochot = round((b[2] - a[2]) / ticks, 4), socket_thermal = round((b[3] - a[3]) / ticks, 4),
				vr_thermal = round((b[4] - a[4]) / ticks, 4), hbm_thermal = round((b[5] - a[5]) / ticks, 4), accumulation_ticks = int(ticks))
		cap = self._read('power1_cap') if self.hw is not None else None
		return dict(sclk_mhz_mean = None if sclk is None else round(sclk, 1), power_w_mean = None if power is None else round(power, 1),
			power_cap_w = None if cap is None else cap / 1e6, ppt_residency = None if res is None else res['ppt'], throttle_residency = res,
			gfx_clk_mhz_end = None if b is None else round(b[7], 1), socket_power_w_end = None if b is None else b[6], hotspot_c_end = None if b is None else b[8], hbm_c_end = None if b is None else b[9],
			hwmon_samples = n_samples, sources = dict(hwmon = self.hw is not None, gpu_metrics = self.dv >= 0), unavailable = self.why or None,
			how = 'sysfs hwmon sampled every ~20 ms over the timed region by a native thread of convasr_amd/libconvasr_smi.so (power1_input, freq1_input; no Python in the loop); throttle residencies = differences of the gpu_metrics accumulators read at its two ends through librocm_smi64; no GPU call')


def predicted_comm(engine, world, step_s):
	"""dist.predicted: what convasr_amd.parallel.predict_exposed_comm expects for this run's buckets, rank count and measured step time."""
	try:
		p = engine.predict(world, step_s * 1e3)
		p['per_bucket'] = p['per_bucket'][-4:]  # (the last buckets to complete: the ones that can be exposed; the full table is in DESIGN section 5)
		return p
	except Exception as e:
		return dict(error = f'{type(e).__name__}: {e}')


def measure(args, device, rank, world, use_dist, dist_info, fence, probe = None):
	"""One workload on this rank: W warm-up steps (+ the untimed priming of the step graphs), K timed steps between two fences, the
	second (event-timed, eager) pass.  Returns (the JSON line as a dict -- rank 0 only, else None --, rank 0's launch sequence of one step)."""
	import torch
	import torch.distributed as dist
	import convasr_amd as ca
	from convasr_amd import _lib
	torch.manual_seed(1)
	torch.cuda.reset_peak_memory_stats(device)
	ca.functional.manual_seed(int(os.environ.get('CONVASR_BENCH_DROPOUT_SEED', '1')) + rank)  # (the override: a measurement hook -- step time depends on the data through the chip's clock management)
	wl = Workload(args, device, rank, world)
	flat = wl.flat
	engine = ca.parallel.DataParallelEngine(wl.model, device = device, force_collectives = use_dist, measure_exposed_comm = True) if use_dist else wl.model
	graphed = bool(args.graph) and (engine is wl.model or engine.capturable)
	ca.functional.enable_side_stream_wgrad(device, bool(args.side_stream))

	# with the side stream on or the step replayed from a graph nothing can be event-timed inside the timed region: every per-kernel duration
	# (the dominant kernel included) then comes from the second pass, which runs eagerly on one stream
	elapsed, audio, flops, last, kt, sequence, step = run_timed(args, wl, engine, world, fence, time_main_kernel = not args.no_kernel_timer and rank == 0 and not args.side_stream and not graphed,
		on_warm = (lambda: flat.loss_scaler.current[7].item()) if flat.loss_scaler is not None else None, probe = probe)
	device_state = probe.summary() if probe is not None else None
	overflows0 = run_timed.warm_value or 0.0  # the scaler's overflow count when the timed region starts (warm-up overflows are not the timed region's)
	exposed = engine.exposed_comm_ms() if use_dist else None
	scaler_info = None
	if flat.loss_scaler is not None:
		st = flat.loss_scaler.current.tolist()
		scaler_info = dict(loss_scale = st[0], clean_steps = int(st[1]), overflowed_steps_in_timed_region = int(st[7] - overflows0), overflowed_steps_total = int(st[7]),
			note = 'apex dynamic loss scaling (2^16, x2 per 2000 clean steps, /2 and skip on overflow); an overflowed step skips only the optimizer update')
	graph_info = None
	if graphed:
		graph_info = dict(enabled = True, linear_capture = wl.stepper.linear, graphs = wl.stepper.captures, replays = wl.stepper.replays, eager_warmup_steps = wl.stepper.eager_steps, extra_untimed_warmup_steps = run_timed.extra_warmup,
			node_kinds = [g.get('node_kinds') for g in list(wl.stepper.graphs.values())[:4]], transition_fence_armed = bool(wl.stepper.non_kernel_nodes),  # (hipGraphNodeGetType over every captured step: kernel nodes only -- train.capture_node_kinds)
			note = 'every timed step is one hipGraphLaunch of the whole iteration (forward, CTC, backward, clip, optimizer), one graph per batch shape; the per-step inputs are copied into the graph\'s static buffers inside the timed region')
	# host time of a step with the GPU drained first (the enqueue never waits for queue space): how far ahead of the GPU the Python side
	# can run.  Eagerly the Wav2Letter step needs ~4 ms of it for 16 ms of GPU work and a JasperNetLarge step ~20 ms for ~44 (1,200 launches:
	# one busy neighbour away from host-bound); replayed from a graph a step is a handful of calls.
	host_ms = []
	for i in range(3):
		torch.cuda.synchronize()
		h0 = time.perf_counter()
		step(args.warmup + args.steps + i)
		host_ms.append((time.perf_counter() - h0) * 1e3)
	fence()
	eager_info = None
	if graphed:
		# the same steps launched eagerly with the weight gradients on the side stream (what --graph off times): a short reference region
		wl.stepper.enabled = False
		ca.functional.enable_side_stream_wgrad(device, True)
		n_e = min(args.steps, 20)  # (the same batches as the timed region's first n_e steps)
		for i in range(2):
			step(args.warmup + args.steps + i)
		fence()
		e0 = time.perf_counter()
		for i in range(n_e):
			step(args.warmup + i)
		fence()
		e_el = time.perf_counter() - e0
		e_idx = [wl.batch_of(args.warmup + i) for i in range(n_e)]
		eager_info = dict(ms_per_step = round(1e3 * e_el / n_e, 3), steps = n_e, value = round(sum(wl.audio[j][0] for j in e_idx) / e_el, 1),
			whole_step_frac = round(sum(wl.flops[j][0] + wl.flops[j][1] for j in e_idx) / e_el / PEAK_BF16_DENSE, 4),
			note = 'the same steps launched kernel by kernel (Python enqueues every launch: ~530 per JasperNetLarge step) with the weight gradients on a side stream, timed right after the graph-replayed region on the same device')
	steps2, kt2, sequence2, calls2 = 0, {}, [], None
	ca.functional.join_side_streams()
	ca.functional.enable_side_stream_wgrad(device, False)  # the event-timed pass below runs every kernel alone on the main stream, eagerly
	if wl.stepper is not None:
		wl.stepper.enabled = False
	if not args.no_kernel_timer:
		steps2 = min(args.steps, 5)
		if rank == 0:
			_lib.timer = _lib.KernelTimer()
		c0 = _lib.calls[0]
		for i in range(steps2):
			step(args.warmup + args.steps + i)
		fence()
		calls2 = (_lib.calls[0] - c0) / steps2
		if rank == 0:
			kt2 = _lib.timer.summary()
			sequence2 = list(_lib.timer.sequence)
			_lib.timer = None
	if not sequence and sequence2:  # (side stream / graphs: the launch order of a step as the eager second pass saw it -- the order a profiler's child run sees with --graph off --side-stream off)
		sequence = sequence2[:len(sequence2) // max(steps2, 1)] * args.steps
	if use_dist:
		# (the parameter checksum: after K identical updates from identical initial replicas every rank must hold the same bits)
		mine = torch.stack([torch.tensor(elapsed, dtype = torch.float64, device = device), torch.tensor(exposed if exposed is not None else float('nan'), dtype = torch.float64, device = device), wl.flat.data.double().sum(), wl.flat.data.double().abs().sum()])
		every = [torch.zeros_like(mine) for _ in range(world)]
		dist.all_gather(every, mine)
		per_rank = [float(t[0]) * 1e3 / args.steps for t in every]
		exposed_all = [float(t[1]) for t in every]
		elapsed = max(float(t[0]) for t in every)  # MAX over ranks
		dist_info.update(per_rank_ms = dict(min = round(min(per_rank), 3), max = round(max(per_rank), 3), mean = round(sum(per_rank) / world, 3), all = [round(v, 3) for v in per_rank]),
			exposed_comm_ms = dict(mean = round(sum(exposed_all) / world, 4), max = round(max(exposed_all), 4),
				how = 'HIP event pair on the main stream per step: backward fully enqueued -> communication stream joined (what the step waits for the gradient exchange beyond its own backward pass), mean over the timed steps'),
			bucket_mib = [round((b['hi'] - b['lo']) * 4 / 2 ** 20, 1) for b in engine.buckets], comm_thread = engine._jobs is not None,
			grad_comm_dtype = str(engine.comm_dtype() or torch.float32).replace('torch.', ''), exchange_mib_per_step = round(engine.exchange_bytes() / 2 ** 20, 1),
			replicas_equal = all(bool(torch.equal(t[2:], every[0][2:])) for t in every),
			predicted = predicted_comm(engine, world, elapsed / args.steps))

	line = None
	if rank == 0:
		# whole-job figures: every rank steps through batches of the same padded size (one bucket per iteration), rank 0's own count x world
		value = world * audio[0] / elapsed
		headline = args.workload == 'wav2letter' and args.batch is None and args.secs is None
		conv_flops_per_s = world * (flops[0] + flops[1]) / elapsed
		roof = roofline_of(args, wl, kt, kt2, steps2, conv_flops_per_s, world, graphed = graphed) if (kt or kt2) else None
		if roof is not None:
			kind = 'hipEventDisableSystemFence' if (os.environ.get('CONVASR_TIMER_EVENTS', 'raw') != 'torch' and _lib._TimingEvent.runtime() is not None) else 'torch.cuda.Event'
			roof['events'] = kind + (' (timing-only HIP events: their record carries no system-scope cache writeback / invalidate; default events cost the timed region 1.5 % around the 34 launches per step, these 0.4 %: profiles/r06_event_kinds.txt)' if kind.startswith('hip') else '')
		metric = 'audio-seconds/sec/node (fwd+bwd+CTC) at bs64x15s' if headline else f'audio-seconds/sec/node (fwd+bwd+CTC), {args.workload}' + ('' if args.batch is None and args.secs is None else ' (TEST-ONLY size)')
		line = dict(metric = metric, value = round(value, 1), unit = 'audio-seconds/sec', n_gpus = world, steps = args.steps, warmup = args.warmup,
			ms_per_step = round(1e3 * elapsed / args.steps, 3), higher_is_better = True, scaling = 'weak', vs_baseline = None, dtype = args.dtype, data = 'synthetic',
			config = dict(workload = wl.name, global_batch = wl.batch * world, parallelism = f'dp{world}', side_stream_wgrad = bool(args.side_stream), step_graphs = graph_info,
				host_enqueue_ms_per_step = round(sorted(host_ms)[1], 2), abi_calls_per_eager_step = None if calls2 is None else round(calls2, 1),
				whole_step_frac = round(conv_flops_per_s / world / (PEAK_F32_MFMA if args.dtype == 'f32' else PEAK_BF16_DENSE), 4), whole_step_tflops = round(conv_flops_per_s / world / 1e12, 1), eager_side_stream = eager_info, device_state = device_state,
				peak_hbm_gib = round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2)),  # (this measure()'s allocations: parameters, arena, activations, workspaces, graph pool -- of the card's 288)
			loss = round(float(last['loss']), 4), loss_scaler = scaler_info, dist = dist_info, roofline = roof, parity = None)
		if args.workload == 'jasper_large':
			line['config'].update(padded_audio_seconds_per_sec = round(world * audio[1] / elapsed, 1), padding_overhead = round(audio[1] / audio[0] - 1, 4),
				gflop_per_padded_audio_s_fwd = round(flops[0] / audio[1] / 1e9, 2), gflop_per_step_fwd_bwd = round((flops[0] + flops[1]) / args.steps / 1e9, 1),
				batch_shapes_in_timed_region = len({tuple(wl.batches[wl.batch_of(args.warmup + i)][0].shape) for i in range(args.steps)}),
				note = 'value counts the utterances\' own durations; the kernels also compute the padded frames (temporal_mask = False, like the reference): whole_step_frac is over the padded FLOPs')
	del engine, last, flat, step
	wl.release()
	torch.cuda.empty_cache()
	return line, sequence


def main(argv = None):
	argv = list(sys.argv[1:] if argv is None else argv)
	args = parse_args(argv)
	if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
		return launch_ranks(args, argv)
	if args.launcher_dry_run:
		return dry_run_rank(args)

	world = int(os.environ.get('WORLD_SIZE', '1'))
	rank = int(os.environ.get('RANK', '0'))
	local_rank = int(os.environ.get('LOCAL_RANK', '0'))
	if args.gpus > 1 and world != args.gpus:
		raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE is {world}')
	use_dist = world > 1 or os.environ.get('CONVASR_FORCE_DIST') == '1'  # the latter: single-rank RCCL smoke test of the DP path
	share_gpu = os.environ.get('CONVASR_SHARE_GPU') == '1'  # test hook for a 1-GPU box: every rank on cuda:0
	affinity = pin_to_gpu_numa_node(0 if share_gpu else local_rank) if use_dist and os.environ.get('CONVASR_NO_PIN') != '1' else None  # before any GPU call

	import torch

	# stdout carries exactly one line, the JSON result of rank 0: native libraries (RCCL prints a version banner through C stdio,
	# flushed at exit, i.e. AFTER a Python print) are pointed at stderr for the duration of the run
	sys.stdout.flush()
	real_stdout = os.dup(1)
	os.dup2(2, 1)

	device = torch.device('cuda', 0 if share_gpu else local_rank)
	torch.cuda.set_device(device)
	dist_info = None
	if use_dist:
		import torch.distributed as dist
		os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
		os.environ.setdefault('MASTER_PORT', '29511')
		os.environ.setdefault('RANK', '0')
		os.environ.setdefault('WORLD_SIZE', '1')
		backend = os.environ.get('CONVASR_DIST_BACKEND', 'nccl')  # (test hook: gloo replaces RCCL, which needs one GPU per rank)
		# a rendezvous or a collective that does not complete within 120 s ends the rank (torch's watchdog aborts the process), the
		# launcher then reports the failure: a hang cannot outlive the driver's patience silently
		limit = datetime.timedelta(seconds = float(os.environ.get('CONVASR_DIST_TIMEOUT', 120)))
		if backend == 'nccl':
			dist.init_process_group('nccl', device_id = device, timeout = limit)
		else:
			dist.init_process_group(backend, timeout = limit)
		rccl = None
		try:
			rccl = '.'.join(str(v) for v in torch.cuda.nccl.version()) if backend == 'nccl' else None
		except Exception:
			pass
		dist_info = dict(backend = dist.get_backend() + (' (RCCL)' if backend == 'nccl' else ''), rccl_version = rccl, world_size = dist.get_world_size(),
			launcher = 'torch.distributed.run' if os.environ.get('TORCHELASTIC_RUN_ID') is not None else 'env', timeout_s = limit.total_seconds(), affinity = affinity)

	import convasr_amd as ca
	from convasr_amd import _lib

	def fence():
		if use_dist:
			dist.barrier()
		torch.cuda.synchronize()

	probe = DeviceProbe(device) if rank == 0 and os.environ.get('CONVASR_NO_PROBE') != '1' else None  # (CONVASR_NO_PROBE=1: A/B hook) host-side sampler of the card's clock / power / PPT residency (sysfs + librocm_smi64 through ctypes: no GPU call)
	line, sequence = measure(args, device, rank, world, use_dist, dist_info, fence, probe)
	if use_dist:
		dist.destroy_process_group()
	if rank == 0:
		# the legs below start child processes / use the host cores (measure() released model, optimizer state and workspaces)
		roof = line['roofline']
		if world == 1 and roof is not None and args.dtype in ('bf16', 'f16') and not args.no_traffic:
			roof['traffic'], roof['traffic_source'] = measure_traffic(args, sequence, args.steps)  # (None + the reason when the counters could not be collected: no stale fallback)
		f16_leg = None
		headline_run = args.workload == 'wav2letter' and args.batch is None and args.secs is None and args.dtype == 'bf16'
		if world == 1 and args.dtype == 'bf16' and not args.no_f16_leg and args.workload == 'wav2letter':
			# the same workload in fp16 (what the reference's apex O1-O3 levels compute in; 2.0e-4 in the CTC loss at 64 x 15 s, above north_star's 1e-4: parity.ctc_loss_rel_err): its own model / arena / optimizer / loss scaler
			# (at least 8 warm-up steps here: apex's dynamic scale starts at 2^16 and halves once per overflowed step until the gradients
			# of this random-data workload fit -- that search belongs to the warm-up, not to the timed region)
			args16 = argparse.Namespace(**dict(vars(args), warmup = max(args.warmup, 8), dtype = 'f16', no_kernel_timer = True))
			try:
				l16, _ = measure(args16, device, rank, world, False, None, lambda: torch.cuda.synchronize(), None)
				sc = l16['loss_scaler']
				f16_leg = dict(f16_value = l16['value'], f16_ms_per_step = l16['ms_per_step'], f16_steps = args.steps, f16_warmup = args16.warmup,
					f16_whole_step_frac = l16['config']['whole_step_frac'], f16_loss_scale = sc['loss_scale'], f16_overflowed_steps_in_timed_region = sc['overflowed_steps_in_timed_region'],
					f16_note = 'second timed region right after the headline, same device, same workload and step count, fp16 storage + MFMA under apex O2 dynamic loss scaling (an overflowed step skips only the optimizer update)')
			except Exception as e:  # (a leg must never cost the headline its line)
				import traceback
				traceback.print_exc(file = sys.stderr)
				f16_leg = dict(f16_error = f'{type(e).__name__}: {e}')
				torch.cuda.synchronize()
				torch.cuda.empty_cache()
		parity_legs = {}
		if world == 1 and headline_run and not args.no_parity_legs:
			# the same workload on the two paths that meet north_star's 1e-4, each its own model / arena / optimizer, right after the headline on the
			# same device: bf16x3 (split-operand convs at MFMA rate) and exact fp32 (v_mfma_f32: ~0.2 s per step, three timed steps)
			for name, steps, warmup, timer_off in (('bf16x3', min(args.steps, 10), 2, False), ('bf16x3f', min(args.steps, 10), 2, False), ('f32', min(args.steps, 3), 1, True)):
				argsp = argparse.Namespace(**dict(vars(args), dtype = name, steps = steps, warmup = warmup, no_kernel_timer = timer_off, side_stream = False, graph = False))
				try:
					lp, _ = measure(argsp, device, rank, world, False, None, lambda: torch.cuda.synchronize(), None)
					peak = PEAK_F32_MFMA if name == 'f32' else PEAK_BF16_DENSE
					parity_legs.update({f'{name}_value': lp['value'], f'{name}_ms_per_step': lp['ms_per_step'], f'{name}_steps': steps, f'{name}_warmup': warmup,
						f'{name}_whole_step_tflops': lp['config']['whole_step_tflops'], f'{name}_whole_step_frac': round(lp['config']['whole_step_tflops'] * 1e12 / peak, 4), f'{name}_peak_tflops': peak / 1e12})
					if name == 'bf16x3':
						rp = lp['roofline'] or {}
						parity_legs.update(bf16x3_roofline = dict(kernel = rp.get('kernel'), frac = rp.get('frac'), executed_mfma_frac = rp.get('executed_mfma_frac'), achieved = rp.get('achieved'), wgrad_frac = (rp.get('wgrad') or {}).get('frac'),
							conv_stack = rp.get('conv_stack'), hbm_kernels = rp.get('hbm_kernels')),
							bf16x3_vs_f32_peak = round(lp['config']['whole_step_tflops'] * 1e12 / PEAK_F32_MFMA, 3),
							bf16x3_note = 'second timed region, same device and workload: fp32 storage, every stride-1 conv as hi*hi + hi*lo + lo*hi on the bf16 matrix pipe (csrc/split3.hip); whole_step_frac prices the ALGORITHMIC FLOPs against the 2.5 PF bf16 peak (the MFMA work executed is ~3x that), bf16x3_vs_f32_peak against the 157.3 TF exact-fp32 MFMA peak')
					elif name == 'bf16x3f':
						rp = lp['roofline'] or {}
						parity_legs.update(bf16x3f_roofline = dict(forward_frac = rp.get('frac'), forward_executed_mfma_frac = rp.get('executed_mfma_frac'), dgrad_one_product = rp.get('dgrad_one_product'), wgrad = rp.get('wgrad'), conv_stack = rp.get('conv_stack')),
							bf16x3f_note = 'the bf16x3 forward -- the same launches, the same CTC loss bit for bit (tests/test_split_operand_gpu.py), i.e. parity.ctc_loss_rel_err["bf16x3"] -- with ONE 16-bit product per gradient in the backward (dy rounded once, x_hi read in place from the saved planes, w_hi): gradients of the bf16 headline\'s accuracy, as under the reference\'s apex O2 (models.py:744-762)')
					else:
						parity_legs.update(f32_note = 'fourth timed region, same device and workload: the exact-fp32 parity path (v_mfma_f32_32x32x2_f32 kernels of conv.hip), priced against the 157.3 TF fp32 MFMA peak')
				except Exception as e:  # (a leg must never cost the headline its line)
					import traceback
					traceback.print_exc(file = sys.stderr)
					parity_legs[f'{name}_error'] = f'{type(e).__name__}: {e}'
					_lib.timer = None
					torch.cuda.synchronize()
					torch.cuda.empty_cache()
		if world == 1 and headline_run and not args.no_jasper_leg:
			# BASELINE configs[4] in the driver's record: `bench.py --workload jasper_large --steps 12 --warmup 3` as a bounded leg (the step's efficiency depends on the mix of bucket lengths: the same 12 batches as the stand-alone line)
			argsj = argparse.Namespace(**dict(vars(args), workload = 'jasper_large', dtype = 'f16', steps = 12, warmup = 3, side_stream = True, no_kernel_timer = False, graph = graph_policy(args.graph_opt, 'jasper_large', args.gpus)))
			# (a leg must never cost the headline its line: a failure -- e.g. a runtime that cannot capture the step -- is recorded, and the replayed leg falls back to the eager step)
			lj, leg_note = None, None
			for attempt in (argsj, argparse.Namespace(**dict(vars(argsj), graph = False))):
				try:
					lj, _ = measure(attempt, device, rank, world, False, None, lambda: torch.cuda.synchronize(), DeviceProbe(device))
					break
				except Exception as e:
					import traceback
					traceback.print_exc(file = sys.stderr)
					leg_note = (leg_note + '; ' if leg_note else '') + ('graph-replayed' if attempt.graph else 'eager') + f' leg failed: {type(e).__name__}: {e}'
					_lib.timer = None
					ca.functional.CAPTURING[0] = False
					torch.cuda.synchronize()
					torch.cuda.empty_cache()
					if not attempt.graph:
						break
			if lj is None:
				line['extra'] = dict(jasper_large = dict(error = leg_note))
			else:
				rj = lj['roofline'] or {}
				# the leg's own figures are those of the FASTER way to run this step on one GPU: launched kernel by kernel with the weight gradients overlapped on a
				# side stream (measured in the same call, right after the replayed region) when that beats the one-chain graph replay -- it does by 1-3 % on one
				# rank; `replayed` keeps the graph's figures (what a host-bound data-parallel rank would run)
				eg = lj['config'].get('eager_side_stream') or {}
				primary = eg if (eg.get('ms_per_step') and eg['ms_per_step'] < lj['ms_per_step']) else None
				line['extra'] = dict(jasper_large = dict(value = (primary or lj)['value'], unit = lj['unit'], ms_per_step = (primary or lj)['ms_per_step'], steps = 12, warmup = 3, dtype = 'f16', whole_step_frac = primary['whole_step_frac'] if primary else lj['config']['whole_step_frac'],
					launched = 'eagerly, weight gradients on a side stream' if primary else 'replayed from HIP graphs', replayed = dict(value = lj['value'], ms_per_step = lj['ms_per_step'], whole_step_frac = lj['config']['whole_step_frac']) if attempt.graph else None,
					dominant_kernel_frac = rj.get('frac'), wgrad_frac = (rj.get('wgrad') or {}).get('frac'), conv_stack_frac = (rj.get('conv_stack') or {}).get('frac'),
					host_enqueue_ms_per_step = lj['config']['host_enqueue_ms_per_step'], eager_side_stream = lj['config']['eager_side_stream'], abi_calls_per_eager_step = lj['config']['abi_calls_per_eager_step'], step_graphs = lj['config']['step_graphs'],
					side_stream_wgrad = True, batch_shapes_in_timed_region = lj['config'].get('batch_shapes_in_timed_region'), padding_overhead = lj['config'].get('padding_overhead'),
					loss_scaler = lj['loss_scaler'], device_state = lj['config']['device_state'], peak_hbm_gib = lj['config'].get('peak_hbm_gib'), workload = lj['config']['workload'],
						note = 'BASELINE configs[4] (JasperNetLarge, 32 x 5-20 s bucketed, fp16, NovoGrad) as a bounded leg of the default line: the same 12 batches as python bench.py --workload jasper_large --steps 12 --warmup 3' + (' -- ' + leg_note if leg_note else '')))
		if world == 1 and not args.no_cpu_baseline:
			ref = {}
			line['cpu_baseline'] = cpu_baseline(keep = ref)
			line['parity'] = gpu_parity(ref, device, ref_full = parity_reference() if headline_run else None)
			line['parity']['headline_dtype'] = args.dtype
		if f16_leg is not None:
			line['parity'] = dict(line['parity'] or {}, **f16_leg)
		if parity_legs:
			line['parity'] = dict(line['parity'] or {}, **parity_legs)
		import ctypes
		ctypes.CDLL(None).fflush(None)
		sys.stdout.flush()
		os.dup2(real_stdout, 1)
		print(json.dumps(line), flush = True)
	return 0


if __name__ == '__main__':
	sys.exit(main())
