"""Upper bound for "fewer LDS fragment reads per MFMA" (e.g. a 128 x 64 per-wave tile: 12 reads per 32 MFMAs instead of 16): diagnostic
builds of conv_v2s.hip that skip half of the X fragment reads (frag1), half of the W fragment reads (frag2) or both (frag3) and compute
on stale registers -- timing only.  Runs scratch/layer_time.py under each library in alternation (child processes, one device)."""
import os, re, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = dict(main = 'libconvasr_hip.so', frag1 = 'libconvasr_hip.frag1.so', frag2 = 'libconvasr_hip.frag2.so', frag3 = 'libconvasr_hip.frag3.so')
res = {}
for rnd in range(2):
	for name, lib in libs.items():
		out = subprocess.run([sys.executable, os.path.join(ROOT, 'scratch', 'layer_time.py')], env = dict(os.environ, CONVASR_HIP_LIB = os.path.join(ROOT, 'convasr_amd', lib), PYTHONPATH = ROOT), capture_output = True, text = True).stdout
		for ln in out.splitlines():
			m = re.match(r'(\S+ k\d+ d\d+): ([\d.]+) us', ln)
			if m: res.setdefault(m.group(1), {}).setdefault(name, []).append(float(m.group(2)))
table = {layer: {n: round(min(v), 1) for n, v in d.items()} for layer, d in res.items()}
for layer, d in table.items():
	print(layer, d, {n: round(v / d['main'], 3) for n, v in d.items()})
json.dump(dict(note = 'us per forward launch (64 x 751 frames, bf16), best of two alternating rounds; frag1 / frag2 / frag3 = half of the X / W / both fragment reads skipped (6 / 6 / 4 ds_read_b128 per 16 MFMAs instead of 8), results garbage, timing only', layers = table), open(os.path.join(ROOT, 'gpurun_out', 'r04_ab_fragreads.json'), 'w'), indent = 1)
