"""Turn the rocprofv3 CSVs under gpurun_out/<tag>_* (scratch/r02_profiles.sh) into the small committed summaries under profiles/."""
import csv, glob as _glob, json, collections, sys, os, shutil
class glob:  # newest match first: a directory may hold the files of several calls
	@staticmethod
	def glob(pat, recursive = False):
		return sorted(_glob.glob(pat, recursive = recursive), key = os.path.getmtime, reverse = True)
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
G = 'gpurun_out'
stats = glob.glob(f'{G}/{tag}_stats/**/*kernel_stats.csv', recursive = True)[0]
rows = list(csv.DictReader(open(stats)))
with open(f'profiles/{tag}_bench_kernel_stats.csv', 'w') as f:
	f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic   (warm-up + timed region + host-probe steps + the second event pass: the number of training steps = Calls of sgd_step_kernel)\n')
	w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
	for r in rows: w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
traffic = {}
for c in ['FETCH_SIZE', 'WRITE_SIZE']:
	f = glob.glob(f'{G}/{tag}_pmc_{c}/**/*counter_collection.csv', recursive = True)[0]
	agg = collections.defaultdict(list)
	for r in csv.DictReader(open(f)): agg[r['Kernel_Name']].append(float(r['Counter_Value']))
	for k, v in agg.items(): traffic.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
out = {}
with open(f'profiles/{tag}_bench_hbm_traffic.csv', 'w') as f:
	f.write('# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-traffic\n')
	f.write('# units: KB per dispatch (mean).  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE reports half of a 16 B/lane streaming read (MI355X_MICROARCH.md, HBM)\n')
	f.write('kernel,dispatches,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_per_launch_corrected\n')
	for k, v in sorted(traffic.items(), key = lambda kv: -(kv[1].get('FETCH_SIZE', (0, 0))[0])):
		fs, n = v.get('FETCH_SIZE', (0, 0)); ws, _ = v.get('WRITE_SIZE', (0, 0))
		hb = (2 * fs + ws) * 1024
		f.write('"%s",%d,%.1f,%.1f,%.4g\n' % (k, n, fs, ws, hb))
		out[k] = dict(dispatches = n, fetch_kb = fs, write_kb = ws, hbm_bytes_per_launch = hb)
conv = {k: v for k, v in out.items() if 'conv1d' in k or 'wgrad' in k}
# both instantiations of the dominant kernel together (what bench.py's roofline.traffic reports)
both = [v for k, v in conv.items() if 'conv1d_igemm_v2s_kernel<unsigned short, unsigned short' in k or 'conv1d_igemm_v2s_kernel<unsigned short, false' in k or 'conv1d_igemm_v2s_kernel<unsigned short, true' in k]
if both:
	n = sum(v['dispatches'] for v in both)
	conv['conv1d_igemm_v2s_kernel<bf16 in, bf16 out> (all launches of both instantiations, incl. the memory-bound decoder dgrad)'] = dict(dispatches = n, fetch_kb = sum(v['fetch_kb'] * v['dispatches'] for v in both) / n, write_kb = sum(v['write_kb'] * v['dispatches'] for v in both) / n, hbm_bytes_per_launch = sum(v['hbm_bytes_per_launch'] * v['dispatches'] for v in both) / n)
json.dump(conv, open(f'profiles/{tag}_conv_traffic.json', 'w'), indent = 1)
# SQ counters + clocks of the MFMA kernels
def pmc(dirname):
	f = glob.glob(f'{G}/{dirname}/**/*counter_collection.csv', recursive = True)[0]
	agg = collections.defaultdict(lambda: collections.defaultdict(list))
	for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
	kt = glob.glob(f'{G}/{dirname}/**/*kernel_trace.csv', recursive = True)[0]
	dur = collections.defaultdict(list)
	for r in csv.DictReader(open(kt)): dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
	return agg, dur
sq, _ = pmc(f'{tag}_pmc_sq')
clk, dur = pmc(f'{tag}_pmc_clk')
summary = {}
for k in sq:
	if not any(s in k for s in ('conv1d_igemm_v2s', 'conv1d_wgrad_v2', 'bn_act_fwd', 'bn_act_bwd_apply')): continue
	m = {c: sum(v) / len(v) for c, v in sq[k].items()}
	w = m['SQ_WAVE_CYCLES']
	e = dict(dispatches = len(sq[k]['SQ_WAVE_CYCLES']), **{c: round(v) for c, v in m.items()})
	e['wait_any_over_wave_cycles'] = round(m['SQ_WAIT_ANY'] / w, 4); e['wait_inst_any_over_wave_cycles'] = round(m['SQ_WAIT_INST_ANY'] / w, 4); e['active_inst_over_wave_cycles'] = round(m['SQ_ACTIVE_INST_ANY'] / w, 4)
	if m.get('SQ_LDS_IDX_ACTIVE'): e['lds_bank_conflict_over_lds_active'] = round(m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'], 4)
	if k in clk:
		g = sum(clk[k]['GRBM_GUI_ACTIVE']) / len(clk[k]['GRBM_GUI_ACTIVE']); d = sum(dur[k]) / len(dur[k]); mf = sum(clk[k]['SQ_VALU_MFMA_BUSY_CYCLES']) / len(clk[k]['SQ_VALU_MFMA_BUSY_CYCLES'])
		e['avg_duration_us_under_pmc'] = round(d / 1e3, 1); e['effective_clock_ghz'] = round(g / 8 / d, 3); e['mfma_busy_over_simd_cycles'] = round(mf / (1024 * g / 8), 4)
	summary[k] = e
json.dump(dict(note = 'rocprofv3 --kernel-trace --pmc (8 SQ counters in one pass; GRBM_GUI_ACTIVE + SQ_BUSY_CYCLES + SQ_VALU_MFMA_BUSY_CYCLES in a second) over bench.py --steps 2 --warmup 1; means per dispatch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles; effective clock = GRBM_GUI_ACTIVE / 8 / duration; mfma_busy_over_simd_cycles = MFMA busy cycles / (1024 SIMDs x kernel cycles)', kernels = summary), open(f'profiles/{tag}_pmc_sq.json', 'w'), indent = 1)
for name in ('bench_line', 'bench_line_f16', 'bench_line_bf16x3', 'launcher_n1', 'plain_n1', 'rccl_world1', 'rccl_world1_graph', 'rccl_world1_config4', 'rccl_world1_config4_graph', 'config4_line', 'config4_line_eager', 'bench_infer'):
	src = f'{G}/{tag}_{name}.json'
	if os.path.exists(src) and os.path.getsize(src) > 0: shutil.copy(src, f'profiles/{tag}_{name}.json')
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:14]: print('%-80s %6s calls  avg %8.1f us  %5.1f%%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
for k, v in conv.items(): print(k[:75], round(v['hbm_bytes_per_launch'] / 1e6, 1), 'MB/launch')
for k, v in summary.items(): print(k[:60], {a: b for a, b in v.items() if 'over' in a or 'clock' in a})

# ---------------------------------------------------------------- BASELINE configs[4] (bench.py --workload jasper_large), round 4 on
c4 = glob.glob(f'{G}/{tag}_config4_stats/**/*kernel_stats.csv', recursive = True)
if c4:
	rows4 = list(csv.DictReader(open(c4[0])))
	with open(f'profiles/{tag}_config4_kernel_stats.csv', 'w') as f:
		f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload jasper_large --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer   (JasperNetLarge, fp16, 32 utterances of 5-20 s per step; the number of training steps = Calls of ng_step_kernel)\n')
		w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
		for r in rows4: w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
	tr = {}
	for c in ['FETCH_SIZE', 'WRITE_SIZE']:
		ff = glob.glob(f'{G}/{tag}_config4_pmc_{c}/**/*counter_collection.csv', recursive = True)
		if not ff: continue
		agg = collections.defaultdict(list)
		for r in csv.DictReader(open(ff[0])): agg[r['Kernel_Name']].append(float(r['Counter_Value']))
		for k, v in agg.items(): tr.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
	out4 = {k: dict(dispatches = v.get('FETCH_SIZE', (0, 0))[1], fetch_kb = v.get('FETCH_SIZE', (0, 0))[0], write_kb = v.get('WRITE_SIZE', (0, 0))[0], hbm_bytes_per_launch = (2 * v.get('FETCH_SIZE', (0, 0))[0] + v.get('WRITE_SIZE', (0, 0))[0]) * 1024) for k, v in tr.items() if any(t in k for t in ('conv1d', 'wgrad', 'bn_act'))}
	json.dump(out4, open(f'profiles/{tag}_config4_traffic.json', 'w'), indent = 1)
	tot4 = sum(float(r['TotalDurationNs']) for r in rows4)
	print('--- configs[4]')
	for r in rows4[:12]: print('%-80s %6s calls  avg %8.1f us  %5.1f%%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot4))
if os.path.exists(f'{G}/{tag}_c4_layers.log'): shutil.copy(f'{G}/{tag}_c4_layers.log', f'profiles/{tag}_config4_layers.txt')


# ---------------------------------------------------------------- the split-operand path (bench.py --dtype bf16x3), round 6 on
x3 = glob.glob(f'{G}/{tag}_x3_stats/**/*kernel_stats.csv', recursive = True)
if x3:
	rows3 = list(csv.DictReader(open(x3[0])))
	with open(f'profiles/{tag}_x3_kernel_stats.csv', 'w') as f:
		f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-jasper-leg   (Wav2Letter full, 64 x 15 s, fp32 storage, split-operand convs: the number of training steps = Calls of sgd_step_kernel)\n')
		w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
		for r in rows3: w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
	tr3 = {}
	for c in ['FETCH_SIZE', 'WRITE_SIZE']:
		ff = glob.glob(f'{G}/{tag}_x3_pmc_{c}/**/*counter_collection.csv', recursive = True)
		if not ff: continue
		agg = collections.defaultdict(list)
		for r in csv.DictReader(open(ff[0])): agg[r['Kernel_Name']].append(float(r['Counter_Value']))
		for k, v in agg.items(): tr3.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
	out3 = {}
	for k, v in tr3.items():
		if not any(t in k for t in ('conv1d', 'wgrad', 'bn_act', 'split3')): continue
		fs, n = v.get('FETCH_SIZE', (0, 0)); ws, _ = v.get('WRITE_SIZE', (0, 0))
		out3[k] = dict(dispatches = n, fetch_kb = fs, write_kb = ws, hbm_bytes_per_launch = (2 * fs + ws) * 1024)
	json.dump(dict(note = 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- bench.py --dtype bf16x3 --steps 2 --warmup 1; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (MI355X_MICROARCH.md, HBM), mean per dispatch', kernels = out3), open(f'profiles/{tag}_x3_traffic.json', 'w'), indent = 1)
	print('--- bf16x3')
	tot3 = sum(float(r['TotalDurationNs']) for r in rows3)
	for r in rows3[:10]: print('%-80s %6s calls  avg %8.1f us  %5.1f%%' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot3))
