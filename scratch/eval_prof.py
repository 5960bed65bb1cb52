import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
d = torch.device('cuda:0')
torch.manual_seed(1)
fe = ca.models.LogFilterBankFrontend(64, 16000, 0.02, 0.01, 'hann_window')
model = ca.models.Wav2Letter(64, [38], frontend = fe, dropout = 0, check_time_dim_padded = False, compute_dtype = torch.bfloat16).to(d)
B, secs = 64, 15
x = (torch.rand(B, 16000 * secs) * 2 - 1).to(d); xlen = torch.ones(B, device = d)
model.train()
with torch.no_grad(): model(x, xlen)
model.eval(); model.fuse_conv_bn_eval()
for _ in range(4):
    with torch.no_grad(): model(x, xlen)
torch.cuda.synchronize()
