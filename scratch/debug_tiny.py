import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convasr_amd as ca
from oracle import convasr_oracle as O
g = np.load('tests/golden/tiny_e2e.npz')
T_ = lambda a: torch.as_tensor(np.asarray(a))
d = torch.device('cuda:0')
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
model = ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, frontend = fe, check_time_dim_padded = False, nonlinearity = ('hardtanh', 0, 20), dilation = 2)
sd = {k[3:]: T_(g[k]).clone() for k in g.files if k.startswith('sd/')}
model.load_state_dict(sd, strict=False)
model.to(d).train()
flat = ca.train.FlatParameters(model)
opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
plan = O.jasper_plan(64, [38], nonlinearity = ('hardtanh', 0, 20), dilation = 2, **O.TINY)
wav, xlen, y, ylen = (T_(g[k]) for k in ['wav', 'xlen', 'y', 'ylen'])
bufs = {}
for it in range(2):
    out = model(wav.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
    loss = (out['loss'] * ylen[:, 0].to(d)).mean()
    loss.backward()
    flat.finalize_grads()
    mine = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.requires_grad}
    # oracle, unclipped grads
    for k in sd: sd[k] = sd[k].detach()
    names = [k for k, v in sd.items() if v.is_floating_point() and not k.startswith('frontend.') and 'running_' not in k]
    for k in names: sd[k].requires_grad_(True); sd[k].grad = None
    o = O.jasper_forward(sd, plan, wav, xlen, y, ylen, frontend = dict(nfft = 512, hop_length = 160), training = True)
    lo = (o['loss'] * ylen[:, 0]).mean(); lo.backward()
    print('it', it, 'loss', float(loss), float(lo), 'logits err', float((out['logits'][0].cpu() - o['logits']).abs().max()))
    for k in names:
        ref = sd[k].grad
        e = (mine[k] - ref).abs().max().item(); m = ref.abs().max().item()
        print(f'   {k:40s} max|ref| {m:.3e} err {e:.3e} rel {e / m:.2e}')
    # apply identical update on both sides: oracle SGD
    gn = flat.clip_grad_norm_(100.0); opt.step(); opt.zero_grad()
    with torch.no_grad():
        params = [sd[k] for k in names]
        torch.nn.utils.clip_grad_norm_(params, 100.0)
        for k in names:
            p = sd[k]; gg = p.grad.add(p, alpha = 1e-3)
            if k not in bufs: bufs[k] = gg.clone()
            else: bufs[k].mul_(0.9).add_(gg)
            p.add_(bufs[k], alpha = -1e-2)
    for k in names: sd[k].requires_grad_(False)
    st = model.state_dict()
    for k in names[:3] + names[-2:]:
        e = (st[k].cpu() - sd[k]).abs().max().item(); print(f'   after-step {k:40s} err {e:.3e}')
os.makedirs('gpurun_out', exist_ok=True)
torch.save(dict(mine={k: v.cpu() for k, v in model.state_dict().items()}, oracle={k: v.detach() for k, v in sd.items()}), 'gpurun_out/tiny_after2.pt')
