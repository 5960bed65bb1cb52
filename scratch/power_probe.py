"""Socket power of THIS process's GPU while the bench workload runs: finds the card by PCI address, samples hwmon power1_input (and
sclk from pp_dpm_sclk / freq1_input when readable) every ~5 ms from a thread for the duration of `bench.py --steps N`, run as a child."""
import glob, json, os, subprocess, sys, threading, time
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
props = torch.cuda.get_device_properties(0)
bus = '%04x:%02x:%02x' % (getattr(props, 'pci_domain_id', 0), props.pci_bus_id, props.pci_device_id)
cards = [c for c in glob.glob('/sys/class/drm/card*') if os.path.realpath(c + '/device').split('/')[-1].startswith(bus)]
print('pci', bus, 'cards', cards, flush = True)
card = cards[0]
hw = glob.glob(card + '/device/hwmon/hwmon*')[0]
def rd(p):
	try: return open(p).read().strip()
	except Exception as e: return None
print({k: rd(f'{hw}/{k}') for k in ('power1_cap', 'power1_cap_max', 'power1_input', 'power1_average', 'freq1_input', 'freq2_input', 'temp1_input')}, flush = True)
samples = []
stop = False
def sampler():
	while not stop:
		samples.append((time.time(), int(rd(f'{hw}/power1_input') or 0), int(rd(f'{hw}/freq1_input') or 0)))
		time.sleep(0.005)
th = threading.Thread(target = sampler); th.start()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
t0 = time.time()
out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', str(steps), '--warmup', '5', '--no-cpu-baseline', '--no-traffic', '--no-kernel-timer'] + sys.argv[2:], capture_output = True, text = True)
t1 = time.time()
stop = True; th.join()
line = json.loads(out.stdout.strip().splitlines()[-1])
busy = steps * line['ms_per_step'] / 1e3
win = [s for s in samples if t1 - busy - 0.3 <= s[0] <= t1 - 0.8]   # the timed region ends ~0.5 s before the child exits
pw = sorted(s[1] / 1e6 for s in win); fq = sorted(s[2] / 1e6 for s in win)
res = dict(card = card, cap_w = int(rd(f'{hw}/power1_cap') or 0) / 1e6, ms_per_step = line['ms_per_step'], value = line['value'], dtype = line['dtype'], n_samples = len(win),
	power_w = dict(mean = sum(pw) / max(len(pw), 1), p10 = pw[len(pw) // 10] if pw else None, median = pw[len(pw) // 2] if pw else None, p90 = pw[len(pw) * 9 // 10] if pw else None, max = pw[-1] if pw else None),
	sclk_mhz = dict(median = fq[len(fq) // 2] if fq else None, p10 = fq[len(fq) // 10] if fq else None, p90 = fq[len(fq) * 9 // 10] if fq else None),
	idle_w = sorted(s[1] / 1e6 for s in samples[:50])[25] if len(samples) > 50 else None)
print(json.dumps(res))
json.dump(res, open(os.path.join(root, 'gpurun_out', 'r03_power%s.json' % ('_' + line['dtype'] if line['dtype'] != 'bf16' else '')), 'w'), indent = 1)
