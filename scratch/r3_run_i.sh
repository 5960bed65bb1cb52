mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r03_gpu_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r03_gpu_tests.log | tail -20
bash scratch/r03_profiles.sh 2>&1 | tail -8
python3 -c "
import json
for l in open('gpurun_out/r03_bench_infer.json'): print(l[:400])
"
