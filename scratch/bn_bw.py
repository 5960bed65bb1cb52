"""Microbenchmark of the HBM-bound BN kernels against a plain elementwise torch kernel on the same tensors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0')
def timeit(fn, iters = 20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for C in (768, 256):
    B, T = 64, 751
    y = ops.as_cl(torch.randn(B, C, T, device = d), torch.bfloat16)
    dz = ops.as_cl(torch.randn(B, C, T, device = d), torch.bfloat16)
    z = ops.empty_cl(B, C, T, torch.bfloat16, d)
    sc, sh = torch.rand(C, device = d) + 0.5, torch.randn(C, device = d)
    coef = torch.randn(3 * C, device = d)
    xlen = torch.ones(B, device = d)
    act = ops.act_args(('hardtanh', 0, 20))
    mb = B * T * C * 2 / 1e6
    flat = y.permute(0, 2, 1).reshape(-1)
    out = torch.empty_like(flat)
    rows = []
    rows.append(('torch relu (1R+1W)', timeit(lambda: torch.relu(flat, out = out) if False else torch.clamp(flat, 0, 20, out = out)), 2))
    rows.append(('bn_act p=0', timeit(lambda: ops.bn_act(y, sc, sh, act, xlen = xlen, out = z)), 2))
    rows.append(('bn_act p=0.2', timeit(lambda: ops.bn_act(y, sc, sh, act, xlen = xlen, dropout_p = 0.2, seed = 1, offset = 0, out = z)), 2))
    rows.append(('bwd_apply p=0', timeit(lambda: ops.bn_act_bwd_apply(dz, y, coef, True, sc, sh, act, xlen = xlen, out = z)), 3))
    rows.append(('bwd_apply p=0.2', timeit(lambda: ops.bn_act_bwd_apply(dz, y, coef, True, sc, sh, act, xlen = xlen, dropout_p = 0.2, seed = 1, offset = 0, out = z)), 3))
    for name, ms, streams in rows:
        print(f'C={C} {name:18s} {ms*1e3:7.1f} us  {streams*mb/ms/1e3:6.2f} TB/s')
