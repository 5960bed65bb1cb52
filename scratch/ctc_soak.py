"""Soak test of the polled two-wave CTC sweeps: many launches per configuration, every result compared BITWISE with the first launch of that
configuration (the pipeline has no data-dependent timing in its arithmetic: any difference is a race), losses checked finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops
d = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
bad = 0
for (B, T, C, S) in [(64, 753, 38, 150), (32, 503, 38, 100), (16, 1003, 38, 250), (8, 900, 129, 383), (64, 120, 38, 40), (4, 1100, 38, 511)]:
	torch.manual_seed(S)
	lp = (torch.randn(B, T, C, device = d) * 2).log_softmax(-1).contiguous().transpose(1, 2)
	y = torch.randint(0, C - 1, (B, S), device = d)
	olen = torch.randint(max(T // 2, 2 * S + 1), T + 1, (B, ), device = d)
	ylen = torch.randint(max(S // 2, 1), S + 1, (B, ), device = d)
	nll0, g0 = ops.ctc_loss(lp, y, olen, ylen, C - 1)
	nll0, g0 = nll0.clone(), g0.clone()
	assert torch.isfinite(nll0).all(), (B, T, C, S)
	# a concurrent memory-bound kernel on another stream perturbs the waves' relative timing
	side = torch.cuda.Stream()
	junk = torch.empty(64 << 20, device = d)
	for i in range(n):
		if i % 3 == 0:
			with torch.cuda.stream(side):
				junk.add_(1.0)
		nll, g = ops.ctc_loss(lp, y, olen, ylen, C - 1)
		if not (torch.equal(nll, nll0) and torch.equal(g, g0)):
			bad += 1
	torch.cuda.synchronize()
	print((B, T, C, S), 'launches', n, 'mismatches so far', bad, flush = True)
print('SOAK', 'OK' if bad == 0 else 'FAILED', bad)
