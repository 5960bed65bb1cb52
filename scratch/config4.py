"""BASELINE configs[4] on one GPU: JasperNetBig (dense residuals, 10 blocks x 5 sub-blocks), 32 x 20 s, mixed lengths, bf16, NovoGrad."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
d = torch.device('cuda:0')
torch.manual_seed(1)
name = sys.argv[1] if len(sys.argv) > 1 else 'JasperNetBig'
B, secs = 32, 20
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
model = getattr(ca.models, name)(64, [38], frontend = fe, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d).train()
print(name, 'params', sum(p.numel() for p in model.parameters()) / 1e6, 'M', flush = True)
flat = ca.train.FlatParameters(model); model._convasr_flat = flat
opt = ca.optimizers.NovoGrad(flat, lr = 1e-3, betas = (0.95, 0.5), weight_decay = 1e-3)
g = torch.Generator().manual_seed(2)
x = (torch.rand(B, 16000 * secs, generator = g) * 2 - 1).to(d)
xlen = (0.5 + 0.5 * torch.rand(B, generator = g)).to(d); xlen[0] = 1.0
y = torch.randint(0, 37, (B, 1, 100), generator = g).to(d); ylen = torch.randint(50, 101, (B, 1), generator = g).to(d)
def step(i): return ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = i)
for i in range(2): r = step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5): r = step(2 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(json.dumps(dict(model = name, ms_per_step = round(dt * 1e3, 2), audio_s_per_s = round(float(xlen.sum()) * secs / dt, 1), padded_audio_s_per_s = round(B * secs / dt, 1), loss = round(float(r['loss_cur']), 4), grad_norm = round(float(r['grad_norm']), 3), skipped = bool(r['skipped']))))
