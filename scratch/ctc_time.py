"""Times the CTC kernels alone at the bench shape (B=64, T=751, C=38, targets up to 159 labels) and prints a checksum of nll / grad.
usage: python scratch/ctc_time.py [S_max]"""
import sys, torch
import convasr_amd
from convasr_amd import ops, _lib
S = int(sys.argv[1]) if len(sys.argv) > 1 else 159
torch.manual_seed(0)
B, T, C = 64, 751, 38
lp = torch.randn(B, T, C, device = 'cuda').log_softmax(-1).permute(0, 2, 1)
y = torch.randint(0, C - 1, (B, S), device = 'cuda')
ylen = torch.randint(S // 2, S + 1, (B,), device = 'cuda')
olen = torch.randint(T * 3 // 4, T + 1, (B,), device = 'cuda')
for _ in range(3): nll, g = ops.ctc_loss(lp, y, olen, ylen, C - 1)
for need_grad in (False, True):
	e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
	torch.cuda.synchronize()
	e0.record()
	for _ in range(20): nll, g_ = ops.ctc_loss(lp, y, olen, ylen, C - 1, need_grad = need_grad)
	e1.record()
	torch.cuda.synchronize()
	print('need_grad', need_grad, 'us per call', round(e0.elapsed_time(e1) / 20 * 1000, 1))
print('nll sum', nll.double().sum().item(), 'grad abs sum', g.double().abs().sum().item(), 'hash', nll.view(torch.int32).sum().item(), g.contiguous().view(torch.int32).sum().item())
