import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convasr_amd as ca
g = np.load('tests/golden/tiny_e2e.npz')
T_ = lambda a: torch.as_tensor(np.asarray(a))
d = torch.device('cuda:0')
def run(keep, use_train_step):
    fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
    model = ca.models.JasperNet(64, [38], base_width = 32, kernel_sizes = [11], out_width_factors = [2], dropouts = [0.2], out_width_factors_large = [2, 2], residual = False, repeat = 1, frontend = fe, check_time_dim_padded = False, nonlinearity = ('hardtanh', 0, 20), dilation = 2)
    sd = {k[3:]: T_(g[k]).clone() for k in g.files if k.startswith('sd/')}
    model.load_state_dict(sd, strict=False)
    model.to(d).train()
    flat = ca.train.FlatParameters(model)
    opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3, keep_clipped_grads = keep)
    wav, xlen, y, ylen = (T_(g[k]).to(d) for k in ['wav', 'xlen', 'y', 'ylen'])
    for it in range(2):
        if use_train_step and it == 1:
            r = ca.train.train_step(model, opt, wav, xlen, y, ylen); print('  ts loss', float(r['loss']), float(r['grad_norm']))
        else:
            out = model(wav, xlen, y = y, ylen = ylen)
            loss = (out['loss'] * ylen[:, 0]).mean(); loss.backward()
            gn = flat.clip_grad_norm_(100.0); opt.step(); opt.zero_grad(); print('  loss', float(loss), float(gn))
    st = model.state_dict()
    k = 'backbone.0.conv.0.0.weight'
    print(keep, use_train_step, 'err vs golden', float((st[k].cpu() - T_(g['sd_after2/' + k])).abs().max()))
run(False, False); run(True, False); run(False, True); run(True, True)
