import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convasr_amd as ca
from oracle import convasr_oracle as O
d = torch.device('cuda:0')
gen = torch.Generator().manual_seed(8)
C = 38
for B, T, S in ((2, 2600, 1100), (2, 9000, 3000), (1, 60000, 8000), (64, 753, 150)):
	tg = torch.randint(0, C - 1, (B, S), generator = gen)
	tl, il = torch.full((B, ), S), torch.full((B, ), T)
	lp = torch.randn(T, B, C, generator = gen).log_softmax(-1).to(d)
	ca.ctc.alignment(lp, tg, il, tl, blank = C - 1); torch.cuda.synchronize()
	t0 = time.time(); al = ca.ctc.alignment(lp, tg, il, tl, blank = C - 1); torch.cuda.synchronize(); t1 = time.time()
	print(B, T, S, 'gpu ms', round((t1 - t0) * 1e3, 2), flush = True)
	if T <= 9000:
		for nt in (torch.get_num_threads(), 1):
			torch.set_num_threads(nt)
			t0 = time.time(); ref = O.ctc_alignment(lp.cpu(), tg, il, tl, blank = C - 1); t1 = time.time()
			print('   oracle threads', nt, 's', round(t1 - t0, 2), 'equal', bool(torch.equal(ref, al.cpu())), flush = True)
