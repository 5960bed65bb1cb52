mkdir -p gpurun_out/r3m
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
python bench.py --dtype f32 --steps 3 --warmup 1 --no-traffic --no-cpu-baseline > gpurun_out/r3m/bench_f32.json 2> gpurun_out/r3m/bench_f32.err; echo "f32 rc $?"; python -c "import json;j=json.load(open('gpurun_out/r3m/bench_f32.json'));print('f32', j['ms_per_step'], j['value'], j['roofline']['frac'], j['loss'])"
python bench.py --side-stream --no-traffic --no-cpu-baseline > gpurun_out/r3m/bench_side.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3m/bench_side.json'));print('side stream', j['ms_per_step'], j['value'])"
python bench.py --no-traffic --no-cpu-baseline > gpurun_out/r3m/bench_plain.json 2> /dev/null; python -c "import json;j=json.load(open('gpurun_out/r3m/bench_plain.json'));print('plain', j['ms_per_step'], j['value'])"
python bench.py --dtype f16 --no-cpu-baseline > gpurun_out/r3m/bench_f16.json 2> gpurun_out/r3m/bench_f16.err; python -c "import json;j=json.load(open('gpurun_out/r3m/bench_f16.json'));print('f16', j['ms_per_step'], j['value'], j['roofline']['traffic'], j['roofline']['traffic_source'][:120], j['loss_scaler'])"
