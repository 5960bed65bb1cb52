mkdir -p gpurun_out/r3h
python -m pytest tests/test_round3_gpu.py -q -m gpu -p no:cacheprovider -k "grouped or separable" > gpurun_out/r3h/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r3h/tests.log | tail -30
