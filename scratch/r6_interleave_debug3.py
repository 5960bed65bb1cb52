import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import convasr_amd as ca
import test_split_operand_gpu as T
d = torch.device('cuda:0')
shapes = dict(A = (4, 4), B = (3, 5), C = (5, 3))
order = list('AABABBCABCA')
make_opt = lambda flat: ca.optimizers.AdamW(flat, lr = 1e-3, weight_decay = 1e-2)
variant = sys.argv[1] if len(sys.argv) > 1 else 'both'
eager = T._interleaved(ca, d, make_opt, torch.bfloat16, None, False, order, shapes)
if variant == 'drop_eager':
	eager = (eager[0], eager[1], eager[2], None)
	import gc; gc.collect()
graph = T._interleaved(ca, d, make_opt, torch.bfloat16, None, True, order, shapes, max_graphs = 2 if variant != 'mg64' else 64)
bad = [(i, order[i], a, b) for i, (a, b) in enumerate(zip(eager[0], graph[0])) if a != b]
print(variant, 'BAD' if bad else 'OK', bad, 'params equal', bool(torch.equal(eager[1], graph[1])))
