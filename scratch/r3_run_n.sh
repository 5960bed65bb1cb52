python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r03_gpu_tests_b.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r03_gpu_tests_b.log | tail -12
