#!/bin/bash
# usage (on the GPU box, from the repo root): scratch/r5_run.sh <tag> <what...>   -- the round-5 measurement runner (one parametrised script instead of one r5_run_x.sh per call)
#   tests [pytest args]     pytest -m gpu subset
#   bench [bench args]      python bench.py <args>, line -> gpurun_out/<tag>_bench.json
#   ab_graph                eager vs graph, both workloads, same call
tag=$1; shift
what=$1; shift
mkdir -p gpurun_out
case "$what" in
tests) timeout 1500 python -m pytest "$@" 2>&1 | tail -60 > gpurun_out/${tag}_tests.log; cat gpurun_out/${tag}_tests.log ;;
bench) timeout 1500 python bench.py "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -5 gpurun_out/${tag}_bench.err; cat gpurun_out/${tag}_bench.json ;;
ab_graph)
	for w in wav2letter jasper_large; do
		for g in off on off on; do
			timeout 900 python bench.py --workload $w --graph $g --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --no-f16-leg --no-jasper-leg --no-kernel-timer > gpurun_out/${tag}_${w}_${g}.json 2>> gpurun_out/${tag}_ab.err
			python - <<PY
import json
try:
	l = json.load(open('gpurun_out/${tag}_${w}_${g}.json'))
	print('$w graph=$g', l['value'], l['ms_per_step'], l['config'].get('whole_step_frac'), l['config'].get('host_enqueue_ms_per_step'), l['config'].get('step_graphs'), l['config'].get('device_state', {}) and {k: l['config']['device_state'].get(k) for k in ('sclk_mhz_mean', 'power_w_mean', 'ppt_residency')}, l['loss'])
except Exception as e:
	print('$w graph=$g FAILED', e)
PY
		done
	done
	tail -20 gpurun_out/${tag}_ab.err ;;
prof)
	# rocprofv3 kernel statistics of a bench run: scratch/r5_run.sh <tag> prof <steps+warmup+3 total steps> <bench args>
	n=$1; shift
	cd /tmp && export TMPDIR=/tmp
	rm -rf $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof
	timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg --no-jasper-leg > $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err
	cd $GRAFT_REPO_ROOT
	cp $(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
	python scratch/kstat.py gpurun_out/${tag}_prof $n 45
	cat gpurun_out/${tag}_prof_line.json | cut -c1-400
	rm -rf gpurun_out/${tag}_prof ;;
trace)
	# one step's dispatch list: scratch/r5_run.sh <tag> trace <bench args> -> gpurun_out/<tag>_step_trace.json
	cd /tmp && export TMPDIR=/tmp
	rm -rf $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace
	timeout 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg --no-jasper-leg > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_trace.err
	cd $GRAFT_REPO_ROOT
	python scratch/step_trace.py gpurun_out/${tag}_trace gpurun_out/${tag}_step_trace.json "bench.py $*"
	rm -rf gpurun_out/${tag}_trace ;;
esac
