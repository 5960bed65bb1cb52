"""Per-step kernel table from a rocprofv3 --kernel-trace --stats run: scratch/kstat.py DIR STEPS"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive = True)[0]
steps = float(sys.argv[2])
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/step %.3f, launches/step %.0f' % (tot / steps / 1e6, sum(int(r['Calls']) for r in rows) / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
	print('%-86s %6.1f/step avg %8.1f us  %7.3f ms/step' % (r['Name'][:86], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / steps / 1e6))
