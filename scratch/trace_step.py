"""Per-step breakdown of a rocprofv3 --kernel-trace CSV of bench.py: kernel time by name for the last full training step, idle gaps
(main-queue idle time between kernels) and overlap.  usage: python scratch/trace_step.py <dir with *_kernel_trace.csv> [--gaps]"""
import csv, collections, glob, sys, json
def load(d):
	f = glob.glob(d + '/**/*_kernel_trace.csv', recursive = True)[0]
	rows = list(csv.DictReader(open(f)))
	rows.sort(key = lambda r: int(r['Start_Timestamp']))
	return rows
def last_step(rows):
	idx = [i for i, r in enumerate(rows) if 'logmel' in r['Kernel_Name']]
	return rows[idx[-2]:idx[-1]], (int(rows[idx[-1]]['Start_Timestamp']) - int(rows[idx[-2]]['Start_Timestamp'])) / 1e3
def summary(d, gaps = False):
	step, wall = last_step(load(d))
	iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in step)
	cov, (cs, ce) = 0, iv[0]
	for s, e in iv[1:]:
		if s > ce: cov += ce - cs; cs, ce = s, e
		else: ce = max(ce, e)
	cov += ce - cs
	agg = collections.OrderedDict()
	for r in step:
		n = r['Kernel_Name'].split('(')[0][:90]
		a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
	out = dict(step_wall_us = round(wall, 1), kernel_sum_us = round(sum(v[1] for v in agg.values()), 1), covered_us = round(cov / 1e3, 1), idle_us = round(wall - cov / 1e3, 1), launches = len(step), kernels = {n: dict(launches = c, us = round(t, 1)) for n, (c, t) in sorted(agg.items(), key = lambda kv: -kv[1][1])})
	if gaps:
		prev, g = None, []
		for r in step:
			s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
			if prev and s - prev[1] > 3000: g.append(dict(us = round((s - prev[1]) / 1e3, 1), after = prev[2].split('(')[0][:50], before = r['Kernel_Name'].split('(')[0][:50]))
			if prev is None or e > prev[1]: prev = (s, e, r['Kernel_Name'])
		out['gaps_over_3us'] = g
	return out
if __name__ == '__main__':
	print(json.dumps(summary(sys.argv[1], '--gaps' in sys.argv), indent = 1))
