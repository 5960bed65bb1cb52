"""Build libconvasr_hip.so (gfx950) in-tree with hipcc.  Usage: python -m convasr_amd.build [--force]

Measurement hook: `python -m convasr_amd.build --variant NAME -DFOO=1 ...` builds convasr_amd/libconvasr_hip.NAME.so from the same
sources with extra defines (objects under build/NAME/); CONVASR_HIP_LIB=<path> makes convasr_amd load that file instead, so two
builds can be timed against each other inside one process tree on one device (scratch/ab_lib.py)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libconvasr_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-munsafe-fp-atomics', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function', '-Wno-unused-variable']


def sources():
	return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def stale(target, deps):
	if not os.path.exists(target):
		return True
	t = os.path.getmtime(target)
	return any(os.path.getmtime(d) > t for d in deps)


def build(force = False, verbose = True, variant = None, defines = ()):
	srcs = sources()
	headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(os.path.dirname(HERE), 'include', 'convasr_hip.h')]
	objdir = os.path.join(HERE, 'build') if variant is None else os.path.join(HERE, 'build', variant)
	lib = LIB if variant is None else os.path.join(HERE, f'libconvasr_hip.{variant}.so')
	os.makedirs(objdir, exist_ok = True)
	jobs = []
	objs = []
	for s in srcs:
		o = os.path.join(objdir, os.path.basename(s)[:-4] + '.o')
		objs.append(o)
		if force or stale(o, [s] + headers):
			jobs.append([HIPCC, *FLAGS, *defines, '-c', s, '-o', o])

	def run(cmd):
		if verbose:
			print(' '.join(cmd), flush = True)
		r = subprocess.run(cmd, capture_output = True, text = True)
		if r.returncode != 0:
			raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
		if verbose and r.stderr.strip():
			print(r.stderr, file = sys.stderr)

	with ThreadPoolExecutor(max_workers = min(6, max(1, len(jobs)))) as ex:
		list(ex.map(run, jobs))
	if force or jobs or stale(lib, objs):
		run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, *objs])
	return lib


SMI_LIB = os.path.join(HERE, 'libconvasr_smi.so')


def build_smi_probe(force = False, verbose = True):
	"""libconvasr_smi.so: the host-side gpu_metrics reader of bench.py's device-state sampler (csrc/smi_probe.c, plain C over librocm_smi64;
	measurement infrastructure, not part of the product library).  Returns its path, or None when it cannot be built here."""
	src = os.path.join(CSRC, 'smi_probe.c')
	if force or stale(SMI_LIB, [src]):
		cmd = [os.environ.get('CC', 'gcc'), '-O2', '-shared', '-fPIC', '-I/opt/rocm/include', '-o', SMI_LIB, src, '-L/opt/rocm/lib', '-lrocm_smi64', '-lpthread', '-Wl,-rpath,/opt/rocm/lib']
		if verbose:
			print(' '.join(cmd), flush = True)
		r = subprocess.run(cmd, capture_output = True, text = True)
		if r.returncode != 0:
			if verbose:
				print(r.stderr, file = sys.stderr)
			return None
	return SMI_LIB


if __name__ == '__main__':
	variant = sys.argv[sys.argv.index('--variant') + 1] if '--variant' in sys.argv else None
	print(build(force = '--force' in sys.argv or variant is not None, variant = variant, defines = [a for a in sys.argv[1:] if a.startswith('-D')]))
	if variant is None:
		print(build_smi_probe(force = '--force' in sys.argv))
