"""gpurun_out/r06_x3_pmc_{sq,clk} -> profiles/r06_x3_pmc_sq.json (the split-operand step's MFMA kernels: wave-cycle shares, LDS conflicts, effective clock, MFMA busy)."""
import csv, glob, json, collections, os
G = 'gpurun_out'
def pmc(d):
	f = sorted(glob.glob(f'{G}/{d}/**/*counter_collection.csv', recursive = True), key = os.path.getmtime)[-1]
	agg = collections.defaultdict(lambda: collections.defaultdict(list))
	for r in csv.DictReader(open(f)): agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
	kt = sorted(glob.glob(f'{G}/{d}/**/*kernel_trace.csv', recursive = True), key = os.path.getmtime)[-1]
	dur = collections.defaultdict(list)
	for r in csv.DictReader(open(kt)): dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
	return agg, dur
sq, _ = pmc('r06_x3_pmc_sq'); clk, dur = pmc('r06_x3_pmc_clk')
out = {}
for k in sq:
	if not any(s in k for s in ('conv1d_igemm_v2s', 'conv1d_wgrad_v2', 'bn_act_fwd', 'bn_act_bwd', 'split3', 'pack_split3')): continue
	m = {c: sum(v) / len(v) for c, v in sq[k].items()}; w = m['SQ_WAVE_CYCLES']
	e = dict(dispatches = len(sq[k]['SQ_WAVE_CYCLES']), wait_any_over_wave_cycles = round(m['SQ_WAIT_ANY'] / w, 4), wait_inst_any_over_wave_cycles = round(m['SQ_WAIT_INST_ANY'] / w, 4), active_inst_over_wave_cycles = round(m['SQ_ACTIVE_INST_ANY'] / w, 4))
	if m.get('SQ_LDS_IDX_ACTIVE'): e['lds_bank_conflict_over_lds_active'] = round(m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE'], 4)
	if k in clk:
		g = sum(clk[k]['GRBM_GUI_ACTIVE']) / len(clk[k]['GRBM_GUI_ACTIVE']); d = sum(dur[k]) / len(dur[k]); mf = sum(clk[k]['SQ_VALU_MFMA_BUSY_CYCLES']) / len(clk[k]['SQ_VALU_MFMA_BUSY_CYCLES'])
		e.update(avg_duration_us_under_pmc = round(d / 1e3, 1), effective_clock_ghz = round(g / 8 / d, 3), mfma_busy_over_simd_cycles = round(mf / (1024 * g / 8), 4))
	out[k] = e
json.dump(dict(note = 'rocprofv3 --kernel-trace --pmc (8 SQ counters in one pass; GRBM_GUI_ACTIVE + SQ_BUSY_CYCLES + SQ_VALU_MFMA_BUSY_CYCLES in a second) over bench.py --dtype bf16x3 --steps 2 --warmup 1; means per dispatch; effective clock = GRBM_GUI_ACTIVE / 8 / duration; mfma_busy_over_simd_cycles = MFMA busy cycles / (1024 SIMDs x kernel cycles)', kernels = out), open('profiles/r06_x3_pmc_sq.json', 'w'), indent = 1)
for k, v in out.items(): print(k[:70], v)
