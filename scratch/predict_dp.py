"""DESIGN section 5's table: predicted exposed communication and scaling of the data-parallel step for BASELINE configs[3] (Wav2Letter,
64 x 15 s per GPU) and configs[4] (JasperNetLarge), N = 2 / 4 / 8, from the engine's own bucket boundaries (CPU only: no GPU, no process
group).  Step times: the single-GPU measurements of profiles/.  Usage: python scratch/predict_dp.py [wav2letter_ms jasper_ms]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convasr_amd as ca
from convasr_amd.parallel import DataParallelEngine

ms = dict(wav2letter = float(sys.argv[1]) if len(sys.argv) > 1 else 16.5, jasper_large = float(sys.argv[2]) if len(sys.argv) > 2 else 42.8)
out = {}
for name, build in (('wav2letter', lambda: ca.models.Wav2Letter(64, [38])), ('jasper_large', lambda: ca.models.JasperNetLarge(64, [38]))):
	model = build().train()
	eng = DataParallelEngine(model)
	rows = {}
	for world in (2, 4, 8):
		for eff in (1.0, 0.7, 0.4):
			p = eng.predict(world, ms[name], efficiency = eff, bytes_per_element = 4)
			rows[f'N{world}_eff{eff}'] = {k: p[k] for k in ('exposed_comm_ms', 'comm_ms_total', 'backward_end_ms', 'predicted_scaling')}
			h = eng.predict(world, ms[name], efficiency = eff, bytes_per_element = 2)  # the 16-bit exchange (round 6: DataParallelEngine(grad_comm_dtype); what apex O2 runs use by default)
			rows[f'N{world}_eff{eff}_16bit_exchange'] = {k: h[k] for k in ('exposed_comm_ms', 'comm_ms_total', 'backward_end_ms', 'predicted_scaling')}
		rows[f'N{world}_buckets'] = eng.predict(world, ms[name])['per_bucket']
	out[name] = dict(step_ms = ms[name], gradient_mib = round(eng.flat.numel * 4 / 2 ** 20, 1), buckets_mib = [round((b['hi'] - b['lo']) * 4 / 2 ** 20, 1) for b in eng.buckets], predictions = rows)
	eng.close()
print(json.dumps(out, indent = 1))
