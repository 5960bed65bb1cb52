R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/cmp_bf16 $R/gpurun_out/cmp_f16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cmp_bf16 -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer > $R/gpurun_out/cmp_bf16.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cmp_f16 -- python3 $R/bench.py --dtype f16 --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer > $R/gpurun_out/cmp_f16.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cmp_bf16b -- python3 $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-kernel-timer > $R/gpurun_out/cmp_bf16b.json 2>/dev/null
cut -c1-200 $R/gpurun_out/cmp_bf16.json; cut -c1-200 $R/gpurun_out/cmp_f16.json; cut -c1-200 $R/gpurun_out/cmp_bf16b.json
