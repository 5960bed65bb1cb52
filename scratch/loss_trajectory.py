"""40-step loss trajectory of ONE fixed synthetic batch (8 x 8 s, dropout 0, SGD lr 1e-2 / momentum 0.9 / wd 1e-3 / clip 100: bench.py's
optimizer) on three paths: MI355X bf16, MI355X exact-fp32, and the fp32 CPU oracle.  Written to profiles/r02_loss_trajectory.json.
Also the bench workload itself (64 x 15 s, dropout 0.2, bf16) for 40 steps, to show what its `loss` field does over time."""
import os, sys, json, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
import bench
from oracle import convasr_oracle as O
d = torch.device('cuda:0')
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
torch.set_num_threads(32)
def gpu_run(dt, batch, secs, dropout, steps):
	torch.manual_seed(1); ca.functional.manual_seed(1)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = dropout, check_time_dim_padded = False, compute_dtype = dt)
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	model.to(d).train()
	flat = ca.train.FlatParameters(model); model._convasr_flat = flat
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	x, xlen, y, ylen = bench.synthetic_batch(d, batch = batch, secs = secs)
	out = []
	for i in range(steps):
		r = ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = i)
		out.append((round(float(r['loss']), 4), round(float(r['grad_norm']), 3)))
	return sd, out
res = dict(note = 'loss = mean over utterances of the CTC NLL (train.py:755), grad_norm before clipping at 100; one fixed batch per column')
sd, res['mi355x_bf16_8x8s'] = gpu_run(torch.bfloat16, 8, 8, 0.0, STEPS)
_, res['mi355x_fp32_8x8s'] = gpu_run(torch.float32, 8, 8, 0.0, STEPS)
plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
x, xlen, y, ylen = bench.synthetic_batch('cpu', batch = 8, secs = 8)
bufs, cpu = {}, []
sdc = {k: v.clone() for k, v in sd.items()}
for i in range(STEPS):
	r = O.train_step(sdc, plan, x, xlen, y, ylen, frontend = dict(nfft = 512, hop_length = 160), momentum_buffers = bufs)
	cpu.append((round(float(r['loss']), 4), round(float(r['grad_norm']), 3)))
res['cpu_oracle_fp32_8x8s'] = cpu
_, res['mi355x_bf16_bench_workload_64x15s_dropout0.2'] = gpu_run(torch.bfloat16, 64, 15, 0.2, STEPS)
_, res['mi355x_fp32_bench_workload_64x15s_dropout0.2'] = gpu_run(torch.float32, 64, 15, 0.2, STEPS)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'r02_loss_trajectory.json'), 'w'), indent = 1)
for k, v in res.items():
	if k != 'note': print(k, [a for a, b in v][::4])
