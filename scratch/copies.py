"""Which kernels surround the small copies / ATen kernels of a step: scratch/copies.py DIR  (rocprofv3 --kernel-trace --memory-copy-trace csv)"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive = True):
	for r in csv.DictReader(open(f)):
		ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'][:70]))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive = True):
	for r in csv.DictReader(open(f)):
		ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', '%s %s bytes' % (r.get('Direction', '?'), r.get('Size', r.get('Bytes', '?')))))
ev.sort()
# last step: from the last logmel kernel on
idx = max(i for i, e in enumerate(ev) if 'logmel' in e[3])
prev_end = ev[idx][0]
for s, e, k, name in ev[idx - 12:]:
	print('%s gap %6.1f us  dur %7.1f us  %s' % (k, (s - prev_end) / 1e3, (e - s) / 1e3, name))
	prev_end = e
