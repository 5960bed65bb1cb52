mkdir -p gpurun_out/r3b
python -m pytest tests/test_models_gpu.py tests/test_round3_gpu.py tests/test_fp16_gpu.py -q -m gpu -s -p no:cacheprovider > gpurun_out/r3b/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  " gpurun_out/r3b/tests.log | tail -30
python scratch/ab_bm.py > gpurun_out/r3b/ab_bm.log 2>&1; echo "ab rc $?"; cat gpurun_out/r3b/ab_bm.log | tail -30
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/prof_plain -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r3b/prof_plain.log 2>&1; echo "prof plain rc $?"
CONVASR_FORCE_DIST=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/prof_dist -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-kernel-timer > $GRAFT_REPO_ROOT/gpurun_out/r3b/prof_dist.log 2>&1; echo "prof dist rc $?"
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/r3b/prof_plain.log | cut -c1-300; tail -1 gpurun_out/r3b/prof_dist.log | cut -c1-300
find gpurun_out/r3b -name "*kernel_trace.csv" -size +30M -delete; du -sh gpurun_out/r3b
