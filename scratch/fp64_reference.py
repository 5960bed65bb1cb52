"""Who is closer to exact arithmetic?  Gradients of Wav2Letter full (4 x 10 s) and JasperNetLarge (2 x 5 s features) from the MI355X
fp32 path and from the fp32 CPU oracle, each against the SAME oracle run in float64."""
import os, sys, json, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
from oracle import convasr_oracle as O
torch.set_num_threads(32)
d = torch.device('cuda:0')
def rel(a, b): return float((a.double().cpu() - b.double()).norm() / b.double().norm())
def case(name, model, plan, x, xlen, y, ylen, frontend, names):
	sd = {k: v.clone() for k, v in model.state_dict().items()}
	kw = dict(frontend = frontend, lr = 0.0, momentum = 0.0, weight_decay = 0.0, max_norm = 1e30)
	r32 = O.train_step({k: v.clone() for k, v in sd.items()}, plan, x, xlen, y, ylen, **kw)
	sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
	r64 = O.train_step(sd64, plan, x.double(), xlen, y, ylen, **kw)  # xlen stays fp32: ceil(frac * T) must pick the same frame counts
	model.to(d).train()
	flat = ca.train.FlatParameters(model)
	out = model(x.to(d), xlen.to(d), y = y.to(d), ylen = ylen.to(d))
	(out['loss'] * ylen.to(d)[:, 0]).mean().backward(); flat.finalize_grads()
	p = dict(model.named_parameters())
	res = dict(logits = dict(mi355x_fp32 = rel(out['logits'][0].detach(), r64['logits']), cpu_fp32 = rel(r32['logits'], r64['logits'])))
	for k in names: res[k] = dict(mi355x_fp32 = rel(p[k].grad, r64['grads'][k]), cpu_fp32 = rel(r32['grads'][k], r64['grads'][k]))
	print(name, json.dumps(res, indent = 1), flush = True)
	return res
torch.manual_seed(1)
m = ca.models.Wav2Letter(64, [38], dropout = 0, check_time_dim_padded = False)
x = torch.rand(4, 160000) * 2 - 1; xlen = torch.linspace(0.5, 1, 4); y = torch.randint(0, 37, (4, 1, 100)); ylen = torch.tensor([[50], [60], [80], [100]])
fc = O.frontend_config(); fsd = O.init_state_dict(O.jasper_plan(64, [38], **O.TINY), frontend = fc)
with torch.no_grad(): x = O.logmel_frontend(x, xlen, fsd['frontend.window'], fsd['frontend.mel.weight'], fsd['frontend.mel.bias'], 512, 160)  # fp32 features feed all three runs
out = {}
out['wav2letter_4x10s'] = case('wav2letter', m, O.jasper_plan(64, [38], **O.WAV2LETTER), x, xlen, y, ylen, None, ['backbone.0.conv.0.0.weight', 'backbone.3.conv.1.0.weight', 'backbone.6.conv.0.0.weight', 'backbone.7.conv.0.0.weight', 'decoder.0.weight'])
torch.manual_seed(3)
m = ca.models.JasperNetLarge(64, [38], dropout = 0, check_time_dim_padded = False)
g = torch.Generator().manual_seed(5)
x = torch.randn(2, 64, 501, generator = g); xlen = torch.tensor([1.0, 0.7]); y = torch.randint(0, 37, (2, 1, 40), generator = g); ylen = torch.tensor([[40], [25]])
out['jasperlarge_2x5s'] = case('jasperlarge', m, O.jasper_plan(64, [38], **O.JASPERNET_LARGE), x, xlen, y, ylen, None, ['backbone.10.conv_residual.0.weight', 'backbone.10.conv.4.0.weight', 'backbone.5.conv_residual.2.weight', 'backbone.0.conv.0.0.weight', 'backbone.1.bn.0.weight'])
json.dump(out, open('gpurun_out/r02_fp64_reference.json', 'w'), indent = 1)
