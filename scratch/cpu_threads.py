"""cpu_baseline leg of bench.py at several thread counts (2 x 15 s, one timed step after a warm-up): how the CPU oracle scales on this host."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from oracle import convasr_oracle as O
plan = O.jasper_plan(64, [38], **O.WAV2LETTER); fe = O.frontend_config()
x, xlen, y, ylen = bench.synthetic_batch('cpu', batch = 2, secs = 15)
out = {}
for n in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
	torch.set_num_threads(n)
	sd = O.init_state_dict(plan, seed = 1, frontend = fe); bufs = {}
	ts = []
	for it in range(2):
		t0 = time.perf_counter(); O.train_step(sd, plan, x, xlen, y, ylen, frontend = fe, momentum_buffers = bufs); ts.append(time.perf_counter() - t0)
	out[n] = round(30 / ts[1], 2)
	print(n, ts, out[n], flush = True)
print(json.dumps(out))
