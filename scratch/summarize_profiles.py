"""Turn the rocprofv3 CSVs under gpurun_out/ into the small committed summaries under profiles/."""
import csv, glob, json, collections, sys, os
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
stats = glob.glob('gpurun_out/prof_r1e/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(stats)))
with open(f'profiles/{tag}_bench_kernel_stats.csv', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline   (7 steps incl. warm-up)\n')
    w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows: w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
traffic = {}
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
    f = glob.glob(f'gpurun_out/pmc_bench_{c}/*/*counter_collection.csv')[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)): agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items(): traffic.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
out = {}
with open(f'profiles/{tag}_bench_hbm_traffic.csv', 'w') as f:
    f.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer\n')
    f.write('# units: KB per dispatch (mean).  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE reports half of a 16 B/lane streaming read (MI355X_MICROARCH.md, HBM)\n')
    f.write('kernel,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_per_launch_corrected\n')
    for k, v in sorted(traffic.items(), key = lambda kv: -(kv[1].get('FETCH_SIZE', (0, 0))[0])):
        fs, n = v.get('FETCH_SIZE', (0, 0)); ws, _ = v.get('WRITE_SIZE', (0, 0))
        hb = (2 * fs + ws) * 1024
        f.write('"%s",%d,%.1f,%.1f,%.4g\n' % (k, n, fs, ws, hb))
        out[k] = dict(dispatches = n, fetch_kb = fs, write_kb = ws, hbm_bytes_per_launch = hb)
json.dump({k: v for k, v in out.items() if 'conv1d' in k}, open(f'profiles/{tag}_conv_traffic.json', 'w'), indent = 1)
for k, v in out.items():
    if 'conv1d' in k: print(k[:70], v)
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]: print('%-80s %6s calls  avg %8.1f us  %5.1f%%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
