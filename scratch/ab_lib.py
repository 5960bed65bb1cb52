"""A/B of two builds of libconvasr_hip.so on ONE device: bench.py runs as a child process alternately with each library
(CONVASR_HIP_LIB), several rounds, medians reported.  usage: ab_lib.py NAME_A NAME_B [rounds] [extra bench args]
NAME = 'main' (libconvasr_hip.so) or a --variant name built by `python -m convasr_amd.build --variant NAME -D...`."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
extra = sys.argv[4:]
lib = lambda n: os.path.join(ROOT, 'convasr_amd', 'libconvasr_hip.so' if n == 'main' else f'libconvasr_hip.{n}.so')
res = {n: [] for n in names}
for r in range(rounds):
	for n in names:
		env = dict(os.environ, CONVASR_HIP_LIB = lib(n))
		out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--no-traffic', '--steps', '20', '--warmup', '5', *extra], env = env, stdout = subprocess.PIPE, stderr = subprocess.DEVNULL, text = True).stdout
		j = json.loads(out.strip().splitlines()[-1])
		roof = j['roofline']
		res[n].append(dict(ms = j['ms_per_step'], conv_us = roof['avg_launch_us'], conv_frac = roof['frac'], wgrad_ms = roof['wgrad']['ms_per_step'], wgrad_frac = roof['wgrad']['frac'], fwd_ms = roof['hbm_kernels']['bn_act_fwd_kernel']['ms_per_step'], apply_ms = roof['hbm_kernels']['bn_act_bwd_apply_kernel']['ms_per_step'], stack_ms = roof['conv_stack']['ms_per_step']))
		print(r, n, res[n][-1], flush = True)
for n in names:
	print(n, {k: round(statistics.median(x[k] for x in res[n]), 4) for k in res[n][0]})
