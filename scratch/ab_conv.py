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
variants = [tuple(v.split('=')) for v in sys.argv[1:]] or [('base', '0'), ('alt', '64')]
for (cin, cout, k, dil) in [(256, 256, 11, 1), (256, 384, 13, 1), (384, 384, 13, 1), (384, 256, 13, 1), (512, 512, 17, 1), (640, 640, 21, 1), (768, 768, 25, 1), (768, 896, 29, 2), (896, 768, 29, 2)]:
    B, T = 64, 751
    x = ops.as_cl(torch.randn(B, cin, T, device=d), dt)
    w = torch.randn(cout, cin, k, device=d) / (cin*k)**0.5
    fwd = ops.pack_weight(w, dt, _lib.PACK_FWD)
    flops = 2.0*B*ops.conv_out_len(T,k,1,dil,dil*k//2)*cout*cin*k
    ref = None
    for rnd in range(2):
        for name, flags in variants:
            _lib.load().convasr_debug_set_conv_v2(1 | (int(flags) << 8))
            y = ops.conv1d(x, fwd, cout, k, 1, dil, dil*k//2)
            if ref is None: ref = y.clone()
            same = torch.equal(ref, y)
            ms = timeit(lambda: ops.conv1d(x, fwd, cout, k, 1, dil, dil*k//2))
            print(f'{cin}->{cout} k{k}: {name:18s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TF/s  same={same}', flush=True)
    _lib.load().convasr_debug_set_conv_v2(1)
