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
variants = [tuple(v.split('=')) for v in sys.argv[1:]]
for (cin, cout, k, dil) in [(768, 768, 11, 1), (512, 640, 11, 1), (768, 896, 29, 2), (256, 256, 11, 1)]:
    B, T = 64, 751
    pad = dil*k//2
    x = ops.as_cl(torch.randn(B, cin, T, device=d), dt)
    Tout = ops.conv_out_len(T,k,1,dil,pad)
    dy = ops.as_cl(torch.randn(B, cout, Tout, device=d), dt)
    dw = torch.empty(cout, cin, k, device=d)
    flops = 2.0*B*Tout*cout*cin*k
    ref = None
    for rnd in range(2):
        for name, flags in variants:
            _lib.load().convasr_debug_set_conv_v2(1 | (int(flags) << 8))
            ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw)
            if ref is None: ref = dw.clone()
            err = float((dw - ref).abs().max() / ref.abs().max())
            ms = timeit(lambda: ops.conv1d_wgrad(x, dy, cout, k, 1, dil, pad, dw))
            print(f'{cin}->{cout} k{k}: {name:12s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s  relerr={err:.1e}', flush=True)
    _lib.load().convasr_debug_set_conv_v2(1)
