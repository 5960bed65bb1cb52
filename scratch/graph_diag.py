"""Eager vs graph-replayed steps of one workload on the same batches: per-step losses (must agree bit for bit) and ms per step, with and
without the weight-gradient side stream.  Usage: python scratch/graph_diag.py [jasper_large|wav2letter] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import convasr_amd as ca

workload = sys.argv[1] if len(sys.argv) > 1 else 'jasper_large'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
out = {}
modes = [(g == '1', s == '1') for g, s in (m.split(',') for m in os.environ.get('DIAG_MODES', '0,0;0,1;1,0;1,1').split(';'))]
for graph, side in modes:
	if True:
		args = bench.parse_args(['--workload', workload, '--graph', 'on', '--steps', str(steps), '--warmup', '2'])
		torch.manual_seed(1)
		ca.functional.manual_seed(1)
		wl = bench.Workload(args, dev, 0, 1)
		ca.functional.enable_side_stream_wgrad(dev, side)
		args.graph = graph
		step = wl.make_stepper(wl.model, 1)
		n = len(wl.batches)
		losses = []
		for i in range(2 * n):  # two visits of every batch: eager warm-up, then capture + replay; synchronised, losses recorded
			r = step(i)
			losses.append((float(r['loss']), float(r['grad_norm'])))
		assert wl.prime_graphs(step, range(2 * n, 3 * n)) == 0
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for i in range(2 * n, 3 * n):
			r = step(i)
		torch.cuda.synchronize()
		ms = (time.perf_counter() - t0) * 1e3 / n
		for i in range(3 * n, 3 * n + 4):
			r = step(i)
			losses.append((float(r['loss']), float(r['grad_norm'])))
		out[f'graph={graph} side={side}'] = dict(ms_per_step = round(ms, 3), losses = losses, graphs = wl.stepper.captures, replays = wl.stepper.replays)
		print(f'graph={graph} side={side}: {ms:.3f} ms/step, graphs {wl.stepper.captures}, replays {wl.stepper.replays}', flush = True)
		print('   losses', [round(l[0], 4) for l in losses], flush = True)
		ca.functional.join_side_streams()
		ca.functional.enable_side_stream_wgrad(dev, False)
		wl.release()
		del wl, step, r
		torch.cuda.empty_cache()
json.dump(out, open(os.path.join(bench.ROOT, 'gpurun_out', f'graph_diag_{workload}.json'), 'w'), indent = 1)
for side in (False, True):
	if f'graph=False side={side}' in out and f'graph=True side={side}' in out:
		a, b = out[f'graph=False side={side}']['losses'], out[f'graph=True side={side}']['losses']
		print(f'side={side}: eager == graph losses:', a == b)
