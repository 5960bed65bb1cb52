"""bench_infer.py -- SURVEY 8(f) row f1: the fused eval path under the reference's online-latency protocol and as batch throughput.

    python bench_infer.py [--model Wav2Letter] [--rps 60] [--duration 5] [-B 1] [-T 6.0] [--no-graph]

Protocol (benchmark_online.py:108-160 of the reference, restated): a pinned host batch of B x T seconds of 16 kHz audio; per
request: host -> device copy, logmel frontend + instance norm + conv stack with batch-norm folded into the packed weights
(fuse_conv_bn_eval) + decoder + log-softmax, logits back to the host, synchronize.  100 warm-up requests, then a uniformly
random arrival schedule of rps x duration requests; latency = completion - scheduled arrival; idle = time spent waiting
for the next arrival.  With HIP graphs (default) the ~45 launches of one forward are captured once and replayed per request.
Also prints the offline throughput of the same path at 64 x 15 s.  One JSON line per measurement; random-init weights,
synthetic audio (there is no network for checkpoints here).  bench.py (training throughput) stays the headline bench.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
SAMPLE_RATE = 16000


def build_model(name, device, dtype):
	import convasr_amd as ca
	fe = ca.models.LogFilterBankFrontend(64, SAMPLE_RATE, 0.02, 0.01, 'hann_window')
	model = getattr(ca.models, name)(64, [38], frontend = fe, check_time_dim_padded = False, compute_dtype = dtype, dict = lambda logits, log_probs, olen, **kwargs: log_probs[0])
	model.to(device)
	# one training-mode forward so the running statistics are not the (degenerate) initial ones, then fold BN into the convs
	model.train()
	with torch.no_grad():
		model(torch.rand(2, SAMPLE_RATE * 2, device = device) * 2 - 1, torch.ones(2, device = device))
	model.eval()
	model.fuse_conv_bn_eval()
	return model


def online(model, device, B, T, rps, duration, warmup, use_graph):
	width = int(math.ceil(T * SAMPLE_RATE / 128) * 128)
	host = (torch.rand(B, width) * 2 - 1).pin_memory()
	xlen = torch.ones(B, device = device)
	static_in = torch.empty(B, width, device = device)

	def forward():
		return model(static_in, xlen)

	torch.set_grad_enabled(False)
	static_in.copy_(host, non_blocking = True)
	out = forward()
	torch.cuda.synchronize()
	graph = None
	if use_graph:
		side = torch.cuda.Stream()
		side.wait_stream(torch.cuda.current_stream())
		with torch.cuda.stream(side):
			for _ in range(3):
				out = forward()
		torch.cuda.current_stream().wait_stream(side)
		torch.cuda.synchronize()
		graph = torch.cuda.CUDAGraph()
		with torch.cuda.graph(graph):
			out = forward()
	host_out = torch.empty(out.shape, dtype = out.dtype).pin_memory()

	def request():
		static_in.copy_(host, non_blocking = True)
		if graph is not None:
			graph.replay()
			res = out
		else:
			res = forward()
		host_out.copy_(res, non_blocking = True)
		torch.cuda.synchronize()

	for _ in range(warmup):
		request()
	n = int(round(duration * rps))
	now = time.perf_counter()
	schedule = (torch.rand(n, dtype = torch.float64).sort()[0] * duration + now).tolist()
	lat, idle = [], 0.0
	for t_req in schedule:
		tic = time.perf_counter()
		if tic < t_req:
			idle += t_req - tic
			time.sleep(t_req - tic)
		request()
		lat.append(time.perf_counter() - t_req)
	# service time without the arrival process: back-to-back requests
	t0 = time.perf_counter()
	for _ in range(50):
		request()
	service = (time.perf_counter() - t0) / 50
	lat = torch.tensor(lat, dtype = torch.float64) * 1e3
	q = lambda p: round(float(lat.quantile(p)), 3)
	return dict(metric = 'online request latency (benchmark_online.py protocol)', unit = 'ms', batch = [B, width], audio_seconds = round(B * width / SAMPLE_RATE, 2), requests = n, rps = rps,
	            hip_graph = graph is not None, mean = round(float(lat.mean()), 3), median = q(0.5), p90 = q(0.9), p95 = q(0.95), p99 = q(0.99), max = round(float(lat.max()), 3),
	            service_ms = round(service * 1e3, 3), real_time_factor = round(service / (B * width / SAMPLE_RATE), 6), idle_fraction = round(idle / duration, 4))


def throughput(model, device, B, secs, iters):
	x = (torch.rand(B, SAMPLE_RATE * secs, device = device) * 2 - 1)
	xlen = torch.ones(B, device = device)
	with torch.no_grad():
		for _ in range(3):
			model(x, xlen)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(iters):
			model(x, xlen)
		torch.cuda.synchronize()
	dt = (time.perf_counter() - t0) / iters
	# roofline of the dominant kernel, HIP events on the launching stream around every launch of it (a second pass of the same batches:
	# an event pair costs ~5 us of stream time); FLOPs = 2 MAC of every forward conv it runs (6.67 GFLOP per audio-second
	# for Wav2Letter full, of which the 38-class decoder's 3.75 GFLOP per batch are memory-bound and booked apart)
	from convasr_amd import _lib
	_lib.timer = _lib.KernelTimer(only = ['conv1d_igemm_v2s_kernel<bf16>'])
	with torch.no_grad():
		for _ in range(iters):
			model(x, xlen)
		torch.cuda.synchronize()
	k = _lib.timer.summary().get('conv1d_igemm_v2s_kernel<bf16>')
	_lib.timer = None
	roof = None
	if k is not None:
		tf = k['work'] / (k['total_ms'] * 1e-3) / 1e12
		roof = dict(bound = 'mfma', kernel = 'conv1d_igemm_v2s_kernel<H, H, 0> (the 18 forward convs per batch, BN folded: bias + activation + mask in the epilogue)', achieved = round(tf, 1), peak = 2500.0, unit = 'TFLOP/s', frac = round(tf / 2500.0, 4), launches_per_batch = k['launches'] // iters, ms_per_batch = round(k['total_ms'] / iters, 3), whole_batch_frac = round(6.67e9 * B * secs / dt / 2.5e15, 4), timing = f'HIP events around every launch of this kernel, {iters} batches right after the timed region')
	return dict(metric = 'offline inference throughput, fused eval path', unit = 'audio-seconds/sec', value = round(B * secs / dt, 1), ms_per_batch = round(dt * 1e3, 3), batch = [B, SAMPLE_RATE * secs], roofline = roof)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument('--model', default = 'Wav2Letter')
	ap.add_argument('--dtype', default = 'bf16', choices = ['bf16', 'f16', 'f32', 'bf16x3', 'f16x3'], help = 'bf16x3 / f16x3: fp32 activations, the convs as split-operand products on the 16-bit matrix pipe (fp32-class logits)')
	ap.add_argument('-B', type = int, default = 1)
	ap.add_argument('-T', type = float, default = 6.0)
	ap.add_argument('--rps', type = float, default = 60)
	ap.add_argument('--duration', type = float, default = 5.0)
	ap.add_argument('--warmup-iterations', type = int, default = 100)
	ap.add_argument('--no-graph', action = 'store_true')
	ap.add_argument('--sample-rate', type = int, default = 16000, help = '8000: the reference\'s published configuration (benchmark_online.py:13-21: JasperNetBig, B = 1 x 6 s at 8 kHz)')
	ap.add_argument('--no-throughput', action = 'store_true')
	args = ap.parse_args()
	global SAMPLE_RATE
	SAMPLE_RATE = args.sample_rate
	device = torch.device('cuda', 0)
	torch.cuda.set_device(device)
	torch.manual_seed(1)
	dtype = dict(bf16 = torch.bfloat16, f16 = torch.float16, f32 = torch.float32).get(args.dtype, torch.float32)
	model = build_model(args.model, device, dtype)
	if args.dtype in ('bf16x3', 'f16x3'):
		model.set_compute_dtype(args.dtype, inference = True)
	common = dict(model = args.model, dtype = args.dtype, sample_rate = SAMPLE_RATE, data = 'synthetic', weights = 'random init')
	if not args.no_throughput:
		print(json.dumps(dict(throughput(model, device, 64, 15, 10), **common)), flush = True)
	if not args.no_graph:
		print(json.dumps(dict(online(model, device, args.B, args.T, args.rps, args.duration, args.warmup_iterations, True), **common)), flush = True)
	print(json.dumps(dict(online(model, device, args.B, args.T, args.rps, args.duration, args.warmup_iterations, False), **common)), flush = True)


if __name__ == '__main__':
	main()
