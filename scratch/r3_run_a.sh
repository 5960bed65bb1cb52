mkdir -p gpurun_out/r3a
python -m pytest tests -q -m gpu -x --deselect tests/test_round3_gpu.py --deselect tests/test_fp16_gpu.py -p no:cacheprovider > gpurun_out/r3a/tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r3a/tests.log
python -m pytest tests/test_round3_gpu.py tests/test_fp16_gpu.py -q -m gpu -s -p no:cacheprovider > gpurun_out/r3a/tests_new.log 2>&1; echo "new rc $?"; grep -E "passed|failed|Error|assert" gpurun_out/r3a/tests_new.log | tail -30
python bench.py --no-traffic > gpurun_out/r3a/bench_bf16.json 2> gpurun_out/r3a/bench_bf16.err; echo "bench rc $?"; cut -c1-400 gpurun_out/r3a/bench_bf16.json
python bench.py --dtype f16 --no-traffic --no-cpu-baseline > gpurun_out/r3a/bench_f16.json 2> gpurun_out/r3a/bench_f16.err; echo "bench f16 rc $?"; cut -c1-400 gpurun_out/r3a/bench_f16.json
