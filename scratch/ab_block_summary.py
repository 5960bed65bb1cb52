"""Joins scratch/ab_block.py's timing table with the FETCH_SIZE pass (dispatch order: layer-major, shape, 3 launches each)."""
import csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = json.load(open(os.path.join(root, 'gpurun_out', 'r04_ab_block_time.json')))
f = glob.glob(os.path.join(root, 'gpurun_out', 'r4f', 'ab_block_pmc', '**', '*counter_collection.csv'), recursive = True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'conv1d_igemm_v2s' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
rows.sort(key = lambda r: int(r['Dispatch_Id']))
layers, shapes = list(t['layers']), t['shapes']
assert len(rows) == len(layers) * len(shapes) * 3, (len(rows), len(layers), len(shapes))
i = 0
for L in layers:
	for s in shapes:
		vals = [float(r['Counter_Value']) for r in rows[i:i + 3]]; i += 3
		t['layers'][L][s]['fetch_mb'] = round(2 * sum(vals) / 3 * 1024 / 1e6, 1)  # FETCH_SIZE is in KB and reads half of a wide streaming read on gfx950
	base = t['layers'][L]['16x2']
	for s in shapes:
		c = t['layers'][L][s]
		c['fetch_vs_16x2'] = round(c['fetch_mb'] / base['fetch_mb'], 3)
t['note'] = 'bf16 forward launches with BN statistics, 64 x 751 frames; us = best of two interleaved rounds of 20 launches; fetch_mb = 2 x FETCH_SIZE (KB) x 1024 per launch, mean of 3 launches under rocprofv3 --pmc FETCH_SIZE (counts Infinity-Cache hits too: bytes that missed the XCD L2)'
json.dump(t, open(os.path.join(root, 'profiles', 'r04_ab_block_shape.json'), 'w'), indent = 1)
for L in layers:
	print(L, {s: (c['us'], c['vs_16x2'], c['fetch_mb'], c['fetch_vs_16x2']) for s, c in t['layers'][L].items()})
