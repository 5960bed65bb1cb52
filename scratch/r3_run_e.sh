mkdir -p gpurun_out/r3e
python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/r3e/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r3e/tests.log | tail -30
for i in 1 2; do
python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3e/bench_plain_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3e/bench_plain_$i.json'));print('plain', j['ms_per_step'], j['roofline']['frac'], j['roofline']['conv_stack'])"
CONVASR_NO_BWD_FUSION=1 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3e/bench_nofuse_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3e/bench_nofuse_$i.json'));print('nofuse', j['ms_per_step'], j['roofline']['frac'], j['roofline']['conv_stack'], j['roofline']['hbm_kernels'].get('bn_act_bwd_reduce_kernel'))"
CONVASR_BN_BWD_BLOCKS=1536 CONVASR_NO_BWD_FUSION=1 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3e/bench_nofuse1536_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3e/bench_nofuse1536_$i.json'));print('nofuse 1536 blocks', j['ms_per_step'], j['roofline']['hbm_kernels'].get('bn_act_bwd_reduce_kernel'))"
done
