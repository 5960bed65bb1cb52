mkdir -p gpurun_out/r3f
python scratch/ab_fused.py > gpurun_out/r3f/ab_fused.log 2>&1; echo "ab rc $?"; tail -12 gpurun_out/r3f/ab_fused.log
python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_round2_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r3f/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r3f/tests.log | tail -30
for i in 1 2; do
python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3f/bench_plain_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3f/bench_plain_$i.json'));print('mfma-sums', j['ms_per_step'], j['roofline']['frac'], j['roofline']['conv_stack'])"
CONVASR_NO_GATE_BITS=1 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3f/bench_nogate_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3f/bench_nogate_$i.json'));print('no gates (valu form)', j['ms_per_step'], j['roofline']['frac'], j['roofline']['conv_stack'])"
done
