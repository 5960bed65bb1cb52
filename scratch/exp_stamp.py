import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); dt = torch.bfloat16
cin, cout, k, dil = 768, 768, 11, 1
B, T = 64, 751
x = ops.as_cl(torch.randn(B, cin, T, device=d), dt)
w = torch.randn(cout, cin, k, device=d) / (cin*k)**0.5
fwd = ops.pack_weight(w, dt, _lib.PACK_FWD)
ntiles = B * 3 * (cout // 128)
for flags in (16, 16 | 64):
    stats = torch.zeros(ntiles * 8 * 8 + 2 * cout, dtype=torch.float64, device=d)
    _lib.load().convasr_debug_set_conv_v2(1 | (flags << 8))
    ops.conv1d(x, fwd, cout, k, 1, dil, dil*k//2, stats=stats)
    torch.cuda.synchronize()
    _lib.load().convasr_debug_set_conv_v2(1)
    raw = stats.view(torch.int64)[:ntiles*64].view(ntiles, 8, 8).cpu().numpy().astype(np.float64)
    Q = raw[0,0,5]
    per = raw[:, :, :5] / Q
    print('flags', flags, 'Q', Q, 'per-step cycles (mean over WGs,waves): issue %.0f  reads+mma %.0f  vmcnt-wait %.0f  barrier %.0f  total %.0f' % tuple(per.mean(axis=(0,1))))
    print('   by wave: total', np.round(per[:, :, 4].mean(axis=0)), ' barrier', np.round(per[:, :, 3].mean(axis=0)), 'compute', np.round(per[:, :, 1].mean(axis=0)))
