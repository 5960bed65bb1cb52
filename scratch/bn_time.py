"""Time of bn_act_fwd with / without the gate output at the bench's layer shapes (bf16, 64 x 751 frames, hardtanh, dropout 0.2)."""
import torch, convasr_amd
from convasr_amd import ops, _lib
d = torch.device('cuda:0'); torch.manual_seed(0)
for C in (256, 512, 768, 1024):
	B, T = 64, 751
	y = ops.as_cl(torch.randn(B, C, T, device = d) * 8, torch.bfloat16)
	sc, sh = torch.rand(C, device = d) + 0.5, torch.randn(C, device = d)
	xl = torch.ones(B, device = d)
	gate = torch.zeros(B * T * C // 8, dtype = torch.uint8, device = d)
	out = torch.empty_like(y)
	res = []
	for g in (None, gate):
		for _ in range(3): ops.bn_act(y, sc, sh, (_lib.ACT_HARDTANH, 0.0, 20.0), xlen = xl, dropout_p = 0.2, seed = 1, offset = 3, out = out, gate = g)
		e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
		torch.cuda.synchronize(); e0.record()
		for _ in range(50): ops.bn_act(y, sc, sh, (_lib.ACT_HARDTANH, 0.0, 20.0), xlen = xl, dropout_p = 0.2, seed = 1, offset = 3, out = out, gate = g)
		e1.record(); torch.cuda.synchronize()
		res.append(e0.elapsed_time(e1) / 50 * 1e3)
	print(f'C {C}: no gate {res[0]:.1f} us ({2 * B * T * C * 2 / res[0] / 1e6:.2f} TB/s), gate {res[1]:.1f} us')
