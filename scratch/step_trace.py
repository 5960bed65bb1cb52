"""One training step as the GPU saw it, from a rocprofv3 --kernel-trace run of bench.py: every dispatch between the last two step_begin_kernel
launches (convasr_step_begin is the first launch of every train_step) with its name, queue and start / end, plus the counts that make
"no ATen kernel and no copy inside the step" checkable from the tree.  Usage: python scratch/step_trace.py <rocprof output dir> <out.json> [label]"""
import csv, glob, json, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive = True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key = lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'step_begin_kernel' in r['Kernel_Name']]
assert len(marks) >= 2, 'fewer than two steps in the trace'
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
def short(n):
	n = re.sub(r'\(.*', '', n)
	return n[:90]
out = dict(label = sys.argv[3] if len(sys.argv) > 3 else '', source = 'rocprofv3 --kernel-trace, dispatches between the last two step_begin_kernel launches (one training step)',
	launches = len(step), wall_us = round((int(rows[b]['Start_Timestamp']) - t0) / 1e3, 1), kernel_us = round(sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3, 1),
	aten_launches = sum('at::native' in r['Kernel_Name'] for r in step), copy_launches = sum('copyBuffer' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name'] for r in step),
	queues = sorted({r.get('Queue_Id', '') for r in step}),
	by_kernel = {}, dispatches = [])
for r in step:
	n = short(r['Kernel_Name'])
	d = out['by_kernel'].setdefault(n, dict(launches = 0, us = 0.0))
	d['launches'] += 1
	d['us'] = round(d['us'] + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 2)
	out['dispatches'].append([n, r.get('Queue_Id', ''), round((int(r['Start_Timestamp']) - t0) / 1e3, 2), round((int(r['End_Timestamp']) - t0) / 1e3, 2)])
json.dump(out, open(sys.argv[2], 'w'), indent = 0)
print({k: out[k] for k in ('label', 'launches', 'wall_us', 'kernel_us', 'aten_launches', 'copy_launches', 'queues')})
print([(k, v) for k, v in out['by_kernel'].items() if 'at::native' in k or 'Buffer' in k])
