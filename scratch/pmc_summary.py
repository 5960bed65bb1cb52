"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: mean of each counter per dispatch, plus derived ratios."""
import csv, glob, sys, collections, json
d = sys.argv[1]
f = glob.glob(d + '/**/*counter_collection.csv', recursive = True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
	agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in agg.items():
	if not any(s in k for s in ('conv1d', 'wgrad', 'bn_act')): continue
	m = {c: sum(v) / len(v) for c, v in cs.items()}
	m['dispatches'] = len(next(iter(cs.values())))
	if 'SQ_WAVE_CYCLES' in m:
		w = m['SQ_WAVE_CYCLES']
		for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_BUSY_CYCLES'):
			if c in m: m[c + '/WAVE_CYCLES'] = round(m[c] / w, 4)
	if 'SQ_LDS_BANK_CONFLICT' in m and m.get('SQ_LDS_IDX_ACTIVE'): m['LDS_CONFLICT/LDS_ACTIVE'] = round(m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'], 4)
	out[k[:80]] = {a: (round(b, 1) if isinstance(b, float) and b > 10 else b) for a, b in m.items()}
print(json.dumps(out, indent = 1))
