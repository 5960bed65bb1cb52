#!/bin/bash
# rocprofv3 kernel statistics of the round-4 tree and of this tree on the same device, one call: scratch/prof_rounds.sh <tag> <workload> <steps>
tag=$1; w=$2; n=${3:-10}
export TMPDIR=/tmp
for tree in r04 r05; do
	if [ $tree = r04 ]; then dir=$GRAFT_REPO_ROOT/scratch/_r04_tree; extra=""; else dir=$GRAFT_REPO_ROOT; extra="--no-jasper-leg --graph off"; fi
	rm -rf $GRAFT_REPO_ROOT/gpurun_out/${tag}_${tree}_prof
	(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_${tree}_prof -- python3 $dir/bench.py --workload $w --steps $n --warmup 3 --side-stream off --no-cpu-baseline --no-traffic --no-kernel-timer --no-f16-leg $extra > /dev/null 2>&1)
	cp $(find $GRAFT_REPO_ROOT/gpurun_out/${tag}_${tree}_prof -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/${tag}_${w}_${tree}_kernel_stats.csv
	echo "== $tree"; python $GRAFT_REPO_ROOT/scratch/kstat.py $GRAFT_REPO_ROOT/gpurun_out/${tag}_${tree}_prof $((n + 6)) 28
	rm -rf $GRAFT_REPO_ROOT/gpurun_out/${tag}_${tree}_prof
done
