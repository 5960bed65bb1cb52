mkdir -p gpurun_out/r3g
python -m pytest tests/test_round3_gpu.py tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r3g/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r3g/tests.log | tail -20
for i in 1 2; do
python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3g/bench_plain_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3g/bench_plain_$i.json'));print('rows2', j['ms_per_step'], {k: (v['frac'], v['ms_per_step']) for k, v in j['roofline']['hbm_kernels'].items() if 'bn_' in k})"
CONVASR_HIP_LIB=$PWD/convasr_amd/libconvasr_hip.rows4.so python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3g/bench_rows4_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3g/bench_rows4_$i.json'));print('rows4', j['ms_per_step'], {k: (v['frac'], v['ms_per_step']) for k, v in j['roofline']['hbm_kernels'].items() if 'bn_' in k})"
done
