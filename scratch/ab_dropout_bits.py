"""Upper bound of "dropout keep-bits off bn_act_fwd's VALU path" (VERDICT round 3 / 4, item 4b): the forward BN + activation + dropout + gate
pass of the Wav2Letter step's 18 layers with the dropout hash (p = 0.2) against the same pass without it (p = 0: what the kernel would do if
the keep bits came from somewhere else for free -- it would still have to READ them, B T C / 8 bytes, which this bound ignores).
Usage: python scratch/ab_dropout_bits.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from convasr_amd import ops, _lib
d = torch.device('cuda:0')
B, T = 64, 751
layers = [256] * 3 + [384] * 3 + [512] * 3 + [640] * 3 + [768] * 3 + [896, 1024] + [256]  # the 18 layers' output channels (the prologue first: 256)
act = (_lib.ACT_HARDTANH, 0.0, 20.0)
def run(p):
	tot = 0.0
	rows = []
	for C in sorted(set(layers)):
		y = ops.as_cl(torch.randn(B, C, T, device = d), torch.bfloat16)
		sc, sh = torch.rand(C, device = d) + 0.5, torch.randn(C, device = d)
		gate = torch.empty(B * T * C // 8, dtype = torch.uint8, device = d)
		out = ops.empty_cl(B, C, T, torch.bfloat16, d)
		f = lambda: ops.bn_act(y, sc, sh, act, dropout_p = p, seed = 1, offset = 0, out = out, gate = gate)
		f(); torch.cuda.synchronize()
		e0, e1 = torch.cuda.Event(enable_timing = True), torch.cuda.Event(enable_timing = True)
		best = 1e9
		for _ in range(3):
			e0.record()
			for _ in range(20):
				f()
			e1.record(); torch.cuda.synchronize()
			best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
		rows.append((C, round(best, 2)))
		tot += best * layers.count(C)
	return tot, rows
on, rows_on = run(0.2)
off, rows_off = run(0.0)
res = dict(what = 'bn_act_fwd over the 18 layer shapes of the Wav2Letter step (64 x 751 frames, bf16, hardtanh, gates stored), microseconds per launch, best of 3 x 20', with_dropout_hash_us_per_step = round(on, 1), without_us_per_step = round(off, 1),
	upper_bound_saving_us_per_step = round(on - off, 1), per_channel_count_with = rows_on, per_channel_count_without = rows_off)
print(json.dumps(res))
if len(sys.argv) > 1:
	json.dump(res, open(sys.argv[1], 'w'), indent = 1)
