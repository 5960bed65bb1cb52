"""Micro-benchmark of the conv kernels on Wav2Letter layer shapes (bf16): fwd, dgrad, wgrad TFLOP/s via HIP events."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
from convasr_amd import ops, _lib
ap = argparse.ArgumentParser()
ap.add_argument('--layers', default = '768,768,11,1,1;256,256,11,1,1;768,896,29,1,2;64,256,11,2,1;896,1024,1,1,1;512,640,11,1,1')
ap.add_argument('--iters', type = int, default = 5)
ap.add_argument('--B', type = int, default = 64)
ap.add_argument('--T', type = int, default = 751)
ap.add_argument('--what', default = 'fwd,dgrad,wgrad')
ap.add_argument('--dtype', default = 'bf16')
ap.add_argument('--nostats', action = 'store_true')
a = ap.parse_args()
d = torch.device('cuda:0')
dt = torch.bfloat16 if a.dtype == 'bf16' else torch.float32
def timeit(fn, iters):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for spec in a.layers.split(';'):
    cin, cout, k, stride, dil = map(int, spec.split(','))
    T = a.T if stride == 1 else 2 * a.T - 1
    pad = dil * k // 2
    x = ops.as_cl(torch.randn(a.B, cin, T, device = d), dt)
    w = torch.randn(cout, cin, k, device = d) / (cin * k) ** 0.5
    fwd, dgr = ops.pack_weight(w, dt, None)
    Tout = ops.conv_out_len(T, k, stride, dil, pad)
    dy = ops.as_cl(torch.randn(a.B, cout, Tout, device = d), dt)
    dw = torch.empty_like(w)
    flops = 2.0 * a.B * Tout * cout * cin * k
    stats = ops.ConvStats(cout, a.B, Tout, d)
    res = []
    if 'fwd' in a.what:
        ms = timeit(lambda: ops.conv1d(x, fwd, cout, k, stride, dil, pad, stats = None if a.nostats else stats), a.iters); res.append(f'fwd {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s')
    if 'dgrad' in a.what and stride == 1:
        ms = timeit(lambda: ops.conv1d(dy, dgr, cin, k, 1, dil, dil * (k - 1) - pad), a.iters); res.append(f'dgrad {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s')
    if 'wgrad' in a.what:
        ms = timeit(lambda: ops.conv1d_wgrad(x, dy, cout, k, stride, dil, pad, dw), a.iters); res.append(f'wgrad {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s')
    print(f'{cin:5d}->{cout:5d} k{k:2d} s{stride} d{dil} T{T:5d} GF {flops/1e9:7.1f} | ' + ' | '.join(res), flush = True)
