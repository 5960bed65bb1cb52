import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from convasr_amd import ops
B, C, T, S = 8, 38, 753, 150
torch.manual_seed(T)
lp = torch.randn(B, C, T).log_softmax(dim = 1)
y = torch.randint(0, C - 1, (B, S)); y[0, : S // 2] = y[0, 0]
olen = torch.randint(max(T // 2, 2 * S + 1), T + 1, (B, )); olen[0] = T
ylen = torch.randint(max(S // 2, 1), S + 1, (B, )); ylen[-1] = S
def aten(dtype):
    l = lp.to(dtype).clone().requires_grad_(True)
    loss = F.ctc_loss(l.permute(2, 0, 1), y, olen, ylen, blank = C - 1, reduction = 'none')
    loss[torch.isfinite(loss)].sum().backward()
    return loss.detach(), l.grad
n64, g64 = aten(torch.float64)
n32, g32 = aten(torch.float32)
nll, grad = ops.ctc_loss(ops.as_cl(lp.cuda()), y, olen, ylen, C - 1)
fin = torch.isfinite(n64)
print('finite', fin.tolist())
print('nll  aten32 vs 64: %.3e   gpu vs 64: %.3e (rel)' % (((n32 - n64).abs() / n64.abs())[fin].max(), ((nll.cpu().double() - n64).abs() / n64.abs())[fin].max()))
print('grad aten32 vs 64: %.3e   gpu vs 64: %.3e   gpu vs aten32: %.3e (max abs)' % ((g32.double() - g64)[fin].abs().max(), (grad.cpu().double() - g64)[fin].abs().max(), (grad.cpu() - g32)[fin].abs().max()))
