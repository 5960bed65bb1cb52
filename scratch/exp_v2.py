import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); dt = torch.bfloat16
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (cin, cout, k, dil) in [(768, 768, 11, 1), (512, 640, 11, 1), (768, 896, 29, 2)]:
    B, T = 64, 751
    x = ops.as_cl(torch.randn(B, cin, T, device=d), dt)
    w = torch.randn(cout, cin, k, device=d) / (cin*k)**0.5
    fwd = ops.pack_weight(w, dt, _lib.PACK_FWD)
    flops = 2.0*B*ops.conv_out_len(T,k,1,dil,dil*k//2)*cout*cin*k
    for flags, name in [(0,'shallow frag pipe'), (64,'deep frag pipe'), (0,'shallow frag pipe'), (64,'deep frag pipe')]:
        _lib.load().convasr_debug_set_conv_v2(1 | (flags << 8))
        ms = timeit(lambda: ops.conv1d(x, fwd, cout, k, 1, dil, dil*k//2))
        print(f'{cin}->{cout} k{k}: {name:22s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s-equiv', flush=True)
    _lib.load().convasr_debug_set_conv_v2(1)
