mkdir -p gpurun_out/r3d
python -m pytest tests -q -m gpu -p no:cacheprovider -x > gpurun_out/r3d/tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|^E  |^FAILED" gpurun_out/r3d/tests.log | tail -30
for i in 1 2; do
python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3d/bench_plain_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3d/bench_plain_$i.json'));print('plain', j['ms_per_step'], j['roofline']['frac'], j['roofline']['conv_stack'])"
CONVASR_NO_BWD_FUSION=1 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3d/bench_nofuse_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3d/bench_nofuse_$i.json'));print('nofuse', j['ms_per_step'], j['roofline']['frac'], j['roofline']['conv_stack'], j['roofline']['hbm_kernels'].get('bn_act_bwd_reduce_kernel'))"
CONVASR_FORCE_DIST=1 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3d/bench_dist_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3d/bench_dist_$i.json'));print('dist thread 64MiB', j['ms_per_step'])"
CONVASR_FORCE_DIST=1 CONVASR_COMM_THREAD=0 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3d/bench_dist_nothread_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3d/bench_dist_nothread_$i.json'));print('dist nothread 64MiB', j['ms_per_step'])"
CONVASR_FORCE_DIST=1 CONVASR_BUCKET_MIB=32 python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3d/bench_dist_32_$i.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3d/bench_dist_32_$i.json'));print('dist thread 32MiB', j['ms_per_step'])"
done
