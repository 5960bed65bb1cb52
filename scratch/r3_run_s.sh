timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r03_s.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r03_s.log | tail -12
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > gpurun_out/r03_s_bench.json 2> gpurun_out/r03_s_bench.err; echo "bench rc $?"; cut -c1-300 gpurun_out/r03_s_bench.json
