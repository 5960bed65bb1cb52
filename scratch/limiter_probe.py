"""What holds the shader clock at ~2.0 GHz while the bench workload runs?  Samples, around `bench.py --steps N` run as a child:
  * every hwmon file of this process's GPU (power1_input / power1_cap, every temp*_input with its label and limits, freq*_input) at ~5 ms;
  * `amd-smi metric --json` (power, clocks, temperatures, and the THROTTLE section: accumulated PPT / socket-thermal / VR-thermal /
    HBM-thermal / PROCHOT residency counters and violation status where the firmware reports them) before, every ~2 s during, after;
  * the raw gpu_metrics table (header + bytes) before / after, for offline parsing.
Output: gpurun_out/r04_limiter.json"""
import base64, glob, json, os, subprocess, sys, threading, time
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
props = torch.cuda.get_device_properties(0)
bus = '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), props.pci_bus_id, props.pci_device_id)
cards = [c for c in glob.glob('/sys/class/drm/card*') if os.path.realpath(c + '/device').split('/')[-1].startswith(bus)]
card = cards[0]
hw = (glob.glob(card + '/device/hwmon/hwmon*') or [None])[0]
def rd(p, binary = False):
	try:
		return open(p, 'rb').read() if binary else open(p).read().strip()
	except Exception as e:
		return None
def hwmon_all():
	return {os.path.basename(f): rd(f) for f in sorted(glob.glob(hw + '/*')) if os.path.isfile(f) and os.path.basename(f) not in ('uevent', )} if hw else {}
def smi(args):
	try:
		r = subprocess.run(args, capture_output = True, text = True, timeout = 30)
		try:
			return json.loads(r.stdout)
		except ValueError:
			return dict(rc = r.returncode, stdout = r.stdout[-4000:], stderr = r.stderr[-1000:])
	except Exception as e:
		return dict(error = repr(e))
def gpu_metrics_raw():
	b = rd(card + '/device/gpu_metrics', binary = True)
	if not b:
		return None
	return dict(structure_size = int.from_bytes(b[0:2], 'little'), format_revision = b[2], content_revision = b[3], base64 = base64.b64encode(b).decode())
out = dict(card = card, pci = bus, hwmon_before = hwmon_all(), gpu_metrics_before = gpu_metrics_raw(),
	amd_smi_before = smi(['amd-smi', 'metric', '--json']), amd_smi_static_limits = smi(['amd-smi', 'static', '--limit', '--json']),
	rocm_smi_before = smi(['rocm-smi', '--showpower', '--showclocks', '--showtemp', '--showvoltage', '--showperflevel', '--showmaxpower', '--json']))
fast, slow, stop = [], [], False
temps = sorted(glob.glob(hw + '/temp*_input')) if hw else []
def fast_sampler():
	while not stop:
		fast.append((time.time(), int(rd(f'{hw}/power1_input') or 0), int(rd(f'{hw}/freq1_input') or 0), [int(rd(t) or 0) for t in temps]))
		time.sleep(0.005)
def slow_sampler():
	while not stop:
		slow.append((time.time(), smi(['amd-smi', 'metric', '--json'])))
		time.sleep(2.0)
ths = [threading.Thread(target = fast_sampler), threading.Thread(target = slow_sampler)]
for t in ths: t.start()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
t0 = time.time()
child = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', str(steps), '--warmup', '5', '--no-cpu-baseline', '--no-traffic', '--no-kernel-timer', '--no-f16-leg'] + sys.argv[2:], capture_output = True, text = True)
t1 = time.time()
stop = True
for t in ths: t.join()
line = json.loads(child.stdout.strip().splitlines()[-1])
busy = steps * line['ms_per_step'] / 1e3
win = [s for s in fast if t1 - busy - 0.3 <= s[0] <= t1 - 0.8]
def pct(v):
	v = sorted(v)
	return dict(mean = sum(v) / len(v), p10 = v[len(v) // 10], median = v[len(v) // 2], p90 = v[len(v) * 9 // 10], max = v[-1]) if v else None
out.update(ms_per_step = line['ms_per_step'], value = line['value'], dtype = line['dtype'], steps = steps, n_fast_samples = len(win),
	power_w = pct([s[1] / 1e6 for s in win]), sclk_mhz = pct([s[2] / 1e6 for s in win]),
	temps_c = {os.path.basename(t): dict(label = rd(t.replace('_input', '_label')), crit = rd(t.replace('_input', '_crit')), emergency = rd(t.replace('_input', '_emergency')), **(pct([s[3][i] / 1e3 for s in win]) or {})) for i, t in enumerate(temps)},
	hwmon_after = hwmon_all(), gpu_metrics_after = gpu_metrics_raw(), amd_smi_after = smi(['amd-smi', 'metric', '--json']),
	amd_smi_during = [dict(t = round(t - t0, 2), in_timed_region = bool(t1 - busy - 0.3 <= t <= t1 - 0.8), metric = m) for t, m in slow])
json.dump(out, open(os.path.join(root, 'gpurun_out', 'r04_limiter.json'), 'w'), indent = 1)
print(json.dumps({k: out[k] for k in ('ms_per_step', 'power_w', 'sclk_mhz', 'temps_c')}))
m = out['amd_smi_after']
print(json.dumps(m)[:3000])
