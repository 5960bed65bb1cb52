"""Time convasr_ctc_loss at the bench shape (64 x 753 frames x 38 classes, 150 labels): HIP events around 20 launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops
d = torch.device('cuda:0')
torch.manual_seed(0)
B, T, C, S = 64, 753, 38, 150
lp = torch.randn(B, T, C, device = d).log_softmax(-1).contiguous().transpose(1, 2)  # (B, C, T) view, channels-last memory
y = torch.randint(0, C - 1, (B, S), device = d)
olen = torch.full((B, ), T, dtype = torch.long, device = d)
ylen = torch.full((B, ), S, dtype = torch.long, device = d)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(3): nll, g = ops.ctc_loss(lp, y, olen, ylen, C - 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
e0.record()
for _ in range(n): nll, g = ops.ctc_loss(lp, y, olen, ylen, C - 1)
e1.record(); torch.cuda.synchronize()
print('ctc fwd+grad us per call', e0.elapsed_time(e1) / n * 1e3, 'nll', float(nll.mean()))
