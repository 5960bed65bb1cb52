"""Uninitialised-read hunt (round 6, the eager / replay interleave of profiles/r06_interleave_race.txt): every torch.empty / empty_like of the package is filled
with a poison pattern (NaN for floats, 0xFF bytes for integers) right after allocation.  A kernel that reads memory nobody wrote turns its result into NaN /
changes it deterministically, instead of depending on what the caching allocator happened to hand out."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import convasr_amd as ca
import test_split_operand_gpu as T

d = torch.device('cuda:0')
shapes = dict(A = (4, 4), B = (3, 5), C = (5, 3))
order = list(sys.argv[2]) if len(sys.argv) > 2 else list('AABABBCABCA')
make_opt = lambda flat: ca.optimizers.AdamW(flat, lr = 1e-3, weight_decay = 1e-2)
mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
_empty, _empty_like = torch.empty, torch.empty_like
POISON = [False]

def poison(t):
	if POISON[0] and t.is_cuda and t.numel() > 0:
		if t.is_floating_point():
			t.fill_(float('nan'))
		elif t.dtype == torch.bool:
			t.fill_(True)
		else:
			t.view(torch.uint8).fill_(0xFF) if t.is_contiguous() else t.fill_(-1)
	return t

torch.empty = lambda *a, **k: poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: poison(_empty_like(*a, **k))

ref = T._interleaved(ca, d, make_opt, torch.bfloat16, None, False, order, shapes)
POISON[0] = True
if os.environ.get('NOFENCE') == '1':
	ca.train.GraphedTrainStep._fence_transition = lambda self, device, eager: None
got = T._interleaved(ca, d, make_opt, torch.bfloat16, None, mode == 'graph', order, shapes, max_graphs = 2)
bad = [(i, order[i], a, b) for i, (a, b) in enumerate(zip(ref[0], got[0])) if a != b]
print(mode, 'poisoned', 'BAD' if bad else 'OK', bad[:4], 'params equal', bool(torch.equal(ref[1], got[1])), 'finite', bool(torch.isfinite(got[1]).all()))
