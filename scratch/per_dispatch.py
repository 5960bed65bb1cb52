"""Per-dispatch durations of the conv kernels of the LAST step in a rocprofv3 kernel trace, in launch order."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive = True)[0]
rows = sorted(csv.DictReader(open(f)), key = lambda r: int(r['Start_Timestamp']))
names = ('conv1d_igemm_v2s', 'conv1d_wgrad_v2', 'wgrad_reduce')
sel = [r for r in rows if any(n in r['Kernel_Name'] for n in names)]
per_step = {'conv1d_igemm_v2s': 33, 'conv1d_wgrad_v2': 17, 'wgrad_reduce': 19}
for n in names:
	rs = [r for r in sel if n in r['Kernel_Name']][-per_step[n]:]
	print(n, [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, 1) for r in rs], 'grid', [int(r['Grid_Size']) // int(r['Workgroup_Size']) if 'Grid_Size' in r else 0 for r in rs][:40])
