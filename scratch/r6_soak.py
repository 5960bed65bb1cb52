"""Soak: 300 steps of the small Wav2Letter (dropout 0.2, three batch shapes in a shuffled order, one shape kept eager by max_graphs = 2) eager vs replayed, per compute type;
loss trajectory and final parameters must be bit-identical (no host fence: captured steps are kernel nodes only)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import convasr_amd as ca
import test_split_operand_gpu as T
d = torch.device('cuda:0')
shapes = dict(A = (4, 4), B = (3, 5), C = (5, 3))
rng = random.Random(5)
order = [rng.choice('ABC') for _ in range(300)]
for name, dt, lvl, make_opt in (('bf16 + AdamW', torch.bfloat16, None, lambda flat: ca.optimizers.AdamW(flat, lr = 1e-4, weight_decay = 1e-2)),
		('fp16 O2 + NovoGrad', torch.float16, 'O2', lambda flat: ca.optimizers.NovoGrad(flat, lr = 1e-4, betas = (0.95, 0.5), weight_decay = 1e-3)),
		('bf16x3f + SGD', 'bf16x3f', None, lambda flat: ca.train.SGD(flat, lr = 1e-4, momentum = 0.9, weight_decay = 1e-3))):
	eager = T._interleaved(ca, d, make_opt, dt, lvl, False, order, shapes)
	graph = T._interleaved(ca, d, make_opt, dt, lvl, True, order, shapes, max_graphs = 2)
	bad = [(i, order[i], a, b) for i, (a, b) in enumerate(zip(eager[0], graph[0])) if a != b]
	st = graph[3]
	print(name, 'OK' if not bad and torch.equal(eager[1], graph[1]) else ('BAD', bad[:3]), 'captures', st.captures, 'replays', st.replays, 'eager', st.eager_steps, 'fence armed', st.non_kernel_nodes, 'last loss', eager[0][-1][0], flush = True)
