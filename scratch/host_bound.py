"""Is the step host-bound?  Enqueue time of a training step (perf_counter around train_step, no synchronisation inside) against its GPU
time (wall with a synchronisation after N steps), for both bench workloads; plus the step's launch count from the kernel timer."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import convasr_amd as ca
from convasr_amd import _lib
out = {}
for workload in ('jasper_large', 'wav2letter'):
	args = bench.parse_args(['--workload', workload, '--steps', '10', '--warmup', '3'])
	d = torch.device('cuda:0'); torch.cuda.set_device(d)
	torch.manual_seed(1); ca.functional.manual_seed(1)
	wl = bench.Workload(args, d, 0, 1)
	def step(i):
		x, xlen, y, ylen = wl.batches[i % len(wl.batches)]
		return ca.train.train_step(wl.model, wl.opt, x, xlen, y, ylen, iteration = i)
	for i in range(3): step(i)
	torch.cuda.synchronize()
	enq = []
	t0 = time.perf_counter()
	for i in range(10):
		a = time.perf_counter(); step(3 + i); enq.append(time.perf_counter() - a)
	t_enq = time.perf_counter() - t0
	torch.cuda.synchronize()
	t_all = time.perf_counter() - t0
	# host alone: the same steps with the GPU drained before each (the enqueue never waits for queue space)
	host = []
	for i in range(5):
		torch.cuda.synchronize(); a = time.perf_counter(); step(13 + i); host.append(time.perf_counter() - a)
	torch.cuda.synchronize()
	_lib.timer = _lib.KernelTimer(only = [])
	step(20); torch.cuda.synchronize()
	n_launch = len(_lib.timer.sequence); _lib.timer = None
	out[workload] = dict(ms_per_step = round(t_all / 10 * 1e3, 2), enqueue_ms_per_step = round(t_enq / 10 * 1e3, 2), host_ms_per_step_gpu_idle = round(sorted(host)[len(host) // 2] * 1e3, 2), timed_launches_per_step = n_launch, per_step_enqueue = [round(e * 1e3, 1) for e in enq])
	print(workload, out[workload], flush = True)
	del wl
	torch.cuda.empty_cache()
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r04_host_bound.json'), 'w'), indent = 1)
