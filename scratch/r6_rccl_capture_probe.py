"""Round 6: which form of an RCCL collective survives hipGraph capture on ROCm 7.2 / torch 2.10 (one rank)?  Each case runs in a child process
(a segfault must not take the others down).  Usage: python scratch/r6_rccl_capture_probe.py -> gpurun_out/r06_rccl_capture_probe.json"""
import json
import os
import subprocess
import sys

CASES = ['main_stream_sync', 'main_stream_async', 'side_stream_async', 'side_stream_sync', 'relaxed_mode_side_async']
CHILD = r'''
import os, sys, json
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
import torch, torch.distributed as dist
case = sys.argv[1]
d = torch.device('cuda:0'); torch.cuda.set_device(d)
dist.init_process_group('nccl', rank = 0, world_size = 1, device_id = d)
x = torch.ones(1 << 20, device = d)
dist.all_reduce(x)   # communicator set up outside the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
kw = dict(capture_error_mode = 'relaxed') if case.startswith('relaxed') else {}
print('capturing', case, flush = True)
with torch.cuda.graph(g, **kw):
	y = x * 2
	if case == 'main_stream_sync':
		dist.all_reduce(y)
	elif case == 'main_stream_async':
		w = dist.all_reduce(y, async_op = True); w.wait()
	else:
		ev = torch.cuda.current_stream().record_event()
		side.wait_event(ev)
		with torch.cuda.stream(side):
			if case == 'side_stream_sync':
				dist.all_reduce(y)
			else:
				w = dist.all_reduce(y, async_op = True); w.wait()
		torch.cuda.current_stream().wait_stream(side)
	z = y + 1
print('captured', flush = True)
x.fill_(3.0)
g.replay(); torch.cuda.synchronize()
print(json.dumps(dict(case = case, ok = True, z0 = float(z[0]))), flush = True)
dist.destroy_process_group()
'''
out = {}
for i, case in enumerate(CASES):
	r = subprocess.run([sys.executable, '-X', 'faulthandler', '-c', CHILD, case, str(29600 + i)], stdout = subprocess.PIPE, stderr = subprocess.PIPE, text = True, timeout = 300)
	lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
	out[case] = json.loads(lines[-1]) if lines else dict(ok = False, returncode = r.returncode, stdout = r.stdout[-300:], stderr = r.stderr[-1500:])
	print(case, out[case], flush = True)
import torch
out['torch'] = torch.__version__
os.makedirs('gpurun_out', exist_ok = True)
json.dump(out, open('gpurun_out/r06_rccl_capture_probe.json', 'w'), indent = 1)
