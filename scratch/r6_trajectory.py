"""Round 6: how long does each compute type's training trajectory stay on the exact-fp32 path's?  The bench workload (Wav2Letter full, 64 x 15 s, SGD lr 1e-2 /
momentum 0.9 / wd 1e-3, dropout 0 so that every type sees the same arithmetic), one fixed batch, 14 steps each, loss per step.  -> gpurun_out/r06_trajectory.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import convasr_amd as ca
d = torch.device('cuda:0')
B, secs, steps = 64, 15, 14
g = torch.Generator().manual_seed(11)
x = (torch.rand(B, 16000 * secs, generator = g) * 2 - 1).to(d)
xlen = torch.linspace(0.5, 1, B).to(d)
y = torch.randint(0, 37, (B, 1, 10 * secs), generator = g).to(d)
ylen = (torch.linspace(0.5, 1, B) * 8 * secs).long().clamp(min = 1).view(B, 1).to(d)
out = {}
for name, dt in (('f32', torch.float32), ('bf16x3', 'bf16x3'), ('bf16x3f', 'bf16x3f'), ('f16x3', 'f16x3'), ('f16x3f', 'f16x3f'), ('f16', torch.float16), ('bf16', torch.bfloat16)):
	torch.manual_seed(1)
	fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
	model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.0, check_time_dim_padded = False, compute_dtype = dt if name not in ('f16', 'f16x3', 'f16x3f') else torch.float32).to(d).train()
	flat = ca.train.FlatParameters(model)
	opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
	if name == 'f16':
		ca.models.data_parallel_and_autocast(model, opt, opt_level = 'O2')
		flat.loss_scaler = ca.train.LossScaler(d, init_scale = 2.0 ** 13)  # (a scale that fits from the first step: no skipped start-up steps in the comparison)
	elif name in ('f16x3', 'f16x3f'):
		ca.models.data_parallel_and_autocast(model, opt, compute_dtype = name)
		flat.loss_scaler = ca.train.LossScaler(d, init_scale = 2.0 ** 13)
	losses = []
	for it in range(steps):
		r = ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = it)
		losses.append(float(r['loss']))
	out[name] = losses
	print(name, ' '.join(f'{v:.4f}' for v in losses), flush = True)
	del model, flat, opt
	torch.cuda.empty_cache()
ref = out['f32']
summary = {k: [abs(a - b) / abs(b) for a, b in zip(v, ref)] for k, v in out.items() if k != 'f32'}
for k, v in summary.items():
	print(k, 'rel. distance from the fp32 trajectory per step:', ' '.join(f'{e:.1e}' for e in v))
json.dump(dict(note = __doc__, losses = out, relative_distance_from_fp32 = summary), open(os.path.join(ROOT, 'gpurun_out', 'r06_trajectory.json'), 'w'), indent = 1)
