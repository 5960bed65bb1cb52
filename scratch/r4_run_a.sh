#!/bin/bash
# round 4, call A: new bench.py (headline incl. fp16 leg), configs[4] bench + kernel stats, world-1 RCCL line with the new dist fields
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4a; mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_line.err
timeout 900 python3 bench.py --workload jasper_large --steps 8 --warmup 3 --no-cpu-baseline > $O/c4_line.json 2> $O/c4_line.err
timeout 600 env CONVASR_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-traffic --no-f16-leg > $O/dist1.json 2> $O/dist1.err
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_stats -- python3 $R/bench.py --workload jasper_large --steps 4 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer > $O/c4_stats_line.json 2> $O/c4_stats.log
cd $R
for f in bench_line c4_line dist1; do tail -c 600 $O/$f.err; echo; head -c 3000 $O/$f.json; echo; done
