#!/bin/bash
# Same-device A/B of the round-4 tree (scratch/_r04_tree, git a4aceec) against this tree: bench.py lines in alternation, one gpurun call.
# usage: scratch/ab_rounds.sh <tag> <workload> <steps>
tag=$1; w=$2; n=${3:-12}
mkdir -p gpurun_out
common="--workload $w --steps $n --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg"
for rep in 1 2; do
	(cd scratch/_r04_tree && timeout 600 python bench.py $common 2>/dev/null | tail -1) > gpurun_out/${tag}_${w}_r04_$rep.json
	timeout 600 python bench.py $common --no-jasper-leg --graph off 2>/dev/null | tail -1 > gpurun_out/${tag}_${w}_r05_eager_$rep.json
	timeout 600 python bench.py $common --no-jasper-leg --graph on 2>/dev/null | tail -1 > gpurun_out/${tag}_${w}_r05_graph_$rep.json
done
python - <<PY
import json, glob
rows = {}
for f in sorted(glob.glob('gpurun_out/${tag}_${w}_*.json')):
	try:
		l = json.load(open(f))
	except Exception as e:
		print(f, 'FAILED', e); continue
	name = f.split('${w}_')[1][:-5]
	rows[name] = dict(value = l['value'], ms_per_step = l['ms_per_step'], host_ms = l['config'].get('host_enqueue_ms_per_step'), frac = (l['config'].get('whole_step_frac') or (l.get('roofline') or {}).get('whole_step_frac')), eager = (l['config'].get('eager_side_stream') or {}).get('ms_per_step'), padded = l['config'].get('padded_audio_seconds_per_sec'))
	print(name, rows[name])
json.dump(dict(what = 'same device, one gpurun call, alternating: round-4 tree (git a4aceec) vs this tree, bench.py $common', rows = rows), open('gpurun_out/${tag}_${w}_summary.json', 'w'), indent = 1)
PY
