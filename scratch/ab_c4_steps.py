"""Same-device, same-process A/B of the round-4 conv changes on the configs[4] step (JasperNetLarge, fp16, bucketed batches): the library's
debug bits switch them off one at a time -- 8192: no conv1x1.hip; 512: K = 1 convs not flattened over the batch; 128: no 128-row last tile;
2048: no whole-launch 128-row tiles.  Each variant runs the same 10 batches twice, in alternation; ms per step = best of the two."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import convasr_amd as ca
from convasr_amd import _lib
workload = sys.argv[1] if len(sys.argv) > 1 else 'jasper_large'
args = bench.parse_args(['--workload', workload, '--steps', '10', '--warmup', '3'])
d = torch.device('cuda:0'); torch.cuda.set_device(d)
torch.manual_seed(1); ca.functional.manual_seed(1)
wl = bench.Workload(args, d, 0, 1)
lib = _lib.load()
def run(bits, n = 10):
	lib.convasr_debug_set_conv_v2(1 | (bits << 8))
	for i in range(3):
		x, xlen, y, ylen = wl.batches[i % len(wl.batches)]
		ca.train.train_step(wl.model, wl.opt, x, xlen, y, ylen, iteration = i)
	torch.cuda.synchronize(); t0 = time.perf_counter()
	for i in range(n):
		x, xlen, y, ylen = wl.batches[(3 + i) % len(wl.batches)]
		ca.train.train_step(wl.model, wl.opt, x, xlen, y, ylen, iteration = 3 + i)
	torch.cuda.synchronize()
	lib.convasr_debug_set_conv_v2(1)
	return (time.perf_counter() - t0) / n * 1e3
VARIANTS = [('shipped', 0), ('no conv1x1 kernel', 8192), ('no conv1x1, K=1 not flattened', 8192 | 512), ('no short last tile', 128), ('none of the round-4 conv changes', 8192 | 512 | 128 | 2048)]
if os.environ.get('AB_ONLY_1X1') == '1':
	VARIANTS = VARIANTS[:2]
res = {}
ROUNDS = int(os.environ.get('AB_ROUNDS', 2))
for rnd in range(ROUNDS):
	for name, bits in VARIANTS:
		res.setdefault(name, []).append(run(bits))
out = {name: dict(ms_per_step = round(min(v), 3), median = round(sorted(v)[len(v) // 2], 3), runs = [round(a, 3) for a in v]) for name, v in res.items()}
base = out['shipped']['ms_per_step']
for name, v in out.items():
	v['vs_shipped'] = round(v['ms_per_step'] / base, 4)
	print(name, v, flush = True)
json.dump(dict(workload = wl.name, note = f'same process, same device, same 10 batches per variant, {ROUNDS} alternating rounds, best (ms_per_step) and median of each', variants = out), open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', os.environ.get('AB_OUT', f'r04_ab_steps_{workload}.json')), 'w'), indent = 1)
