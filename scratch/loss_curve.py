"""Loss trajectory of the bench workload over a few dozen steps (sanity: the bf16 training path keeps learning its fixed batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
import bench
d = torch.device('cuda:0')
torch.manual_seed(1); ca.functional.manual_seed(1)
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0.2, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d).train()
flat = ca.train.FlatParameters(model); model._convasr_flat = flat
opt = ca.train.SGD(flat, lr = 1e-2, momentum = 0.9, weight_decay = 1e-3)
x, xlen, y, ylen = bench.synthetic_batch(d, batch = 16, secs = 8)
out = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    r = ca.train.train_step(model, opt, x, xlen, y, ylen, iteration = i)
    out.append((float(r['loss_cur']), float(r['grad_norm']), bool(r['skipped'])))
for i, (l, g, s) in enumerate(out):
    if i % 4 == 0 or i == len(out) - 1: print(i, round(l, 3), round(g, 2), s)
