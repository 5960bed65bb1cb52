timeout 120 python scratch/ctc_time.py 50
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "ctc" 2>&1 | grep -E "passed|failed|^E  |^FAILED|Error" | tail -5
