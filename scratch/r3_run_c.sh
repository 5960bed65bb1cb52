mkdir -p gpurun_out/r3c
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r3c/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r3c/tests.log | tail -40
python bench.py --no-traffic > gpurun_out/r3c/bench_bf16.json 2> gpurun_out/r3c/bench_bf16.err; echo "bench rc $?"; cut -c1-300 gpurun_out/r3c/bench_bf16.json
CONVASR_NO_PREPACK=1 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3c/bench_bf16_noprepack.json 2> gpurun_out/r3c/bench_bf16_noprepack.err; cut -c1-300 gpurun_out/r3c/bench_bf16_noprepack.json
CONVASR_FORCE_DIST=1 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3c/bench_bf16_dist1.json 2> gpurun_out/r3c/bench_bf16_dist1.err; cut -c1-300 gpurun_out/r3c/bench_bf16_dist1.json
python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3c/bench_bf16_b.json 2> /dev/null; cut -c1-300 gpurun_out/r3c/bench_bf16_b.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_plain -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_plain.log 2>&1; echo "prof plain rc $?"
CONVASR_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_dist -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_dist.log 2>&1; echo "prof dist rc $?"
