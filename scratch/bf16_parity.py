"""Full-size bf16 parity numbers: GPU bf16 / GPU fp32 vs the fp32 oracle and vs the bf16-storage oracle (Wav2Letter full, 4 x 10 s)."""
import os, sys, json, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
from oracle import convasr_oracle as O
torch.manual_seed(1)
torch.set_num_threads(32)
d = torch.device('cuda:0')
FE = dict(nfft = 512, hop_length = 160)
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False)
sd = {k: v.clone() for k, v in model.state_dict().items()}
B, secs = 4, 10
x = torch.rand(B, 16000 * secs) * 2 - 1
xlen = torch.linspace(0.5, 1, B)
y = torch.randint(0, 37, (B, 1, 10 * secs))
ylen = torch.tensor([[50], [60], [80], [100]])
plan = O.jasper_plan(64, [38], **O.WAV2LETTER)
kw = dict(frontend = FE, lr = 0.0, momentum = 0.0, weight_decay = 0.0, max_norm = 1e30)
ref32 = O.train_step(copy.deepcopy(sd), plan, x, xlen, y, ylen, **kw)
ref16 = O.train_step(copy.deepcopy(sd), plan, x, xlen, y, ylen, storage = torch.bfloat16, **kw)
model.to(d).train()
flat = ca.train.FlatParameters(model)
def run(dt):
	model.set_compute_dtype(dt)
	model.load_state_dict(sd)
	flat.zero_grad()
	out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out['loss'] * ylen.to(d)[:, 0]).mean().backward()
	flat.finalize_grads()
	torch.cuda.synchronize()
	return dict(logits = out['logits'][0].detach().cpu().contiguous(), loss_vec = out['loss'].detach().cpu(), grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None})
def cmp(a, b, names):
	rel = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm())
	cos = lambda u, v: float(torch.dot(u.double().flatten(), v.double().flatten()) / (u.double().norm() * v.double().norm()))
	r = dict(logits_rel = rel(a['logits'], b['logits']), loss_rel = float(((a['loss_vec'] - b['loss_vec']).abs() / b['loss_vec'].abs()).max()))
	for k in names: r[k] = (round(cos(a['grads'][k], b['grads'][k]), 6), round(rel(a['grads'][k], b['grads'][k]), 5))
	return r
names = ['backbone.0.conv.0.0.weight', 'backbone.1.conv.1.0.weight', 'backbone.3.conv.1.0.weight', 'backbone.6.conv.0.0.weight', 'backbone.7.conv.0.0.weight', 'decoder.0.weight', 'backbone.5.bn.2.weight', 'backbone.0.bn.0.bias']
g16, g32 = run(torch.bfloat16), run(torch.float32)
res = {'gpu_bf16 vs oracle_bf16_storage': cmp(g16, ref16, names), 'gpu_bf16 vs oracle_fp32': cmp(g16, ref32, names), 'oracle_bf16_storage vs oracle_fp32': cmp(ref16, ref32, names), 'gpu_fp32 vs oracle_fp32': cmp(g32, ref32, names)}
print(json.dumps(res, indent = 1))
