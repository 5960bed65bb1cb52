"""print per-step kernel times from a rocprofv3 kernel_stats.csv dir under gpurun_out/:  python scratch/kstat.py prof_x [steps] [filter ...]"""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7; flt = sys.argv[3:]
import os
rows = list(csv.DictReader(open(max(glob.glob(f'gpurun_out/{d}/*/*kernel_stats.csv'), key = os.path.getmtime))))
tot = 0
for r in rows:
    per = float(r['TotalDurationNs']) / steps / 1e3; tot += per
    if (not flt and per > 20) or any(k in r['Name'] for k in flt): print('%-70s %4d %8.1f us/step avg %7.1f' % (r['Name'][:70], int(r['Calls']) // steps, per, float(r['AverageNs']) / 1e3))
print('total us/step', round(tot, 1))
