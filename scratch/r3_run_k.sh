mkdir -p gpurun_out/r3k
for i in 1 2 3; do
( cd scratch/_r02_tree && python bench.py --no-traffic --no-cpu-baseline 2>/dev/null ) > gpurun_out/r3k/r02_$i.json; python -c "import json;j=json.load(open('gpurun_out/r3k/r02_$i.json'));print('r02 code', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['wgrad']['frac'], j['roofline']['whole_step_frac'])"
python bench.py --no-traffic --no-cpu-baseline 2>/dev/null > gpurun_out/r3k/r03_$i.json; python -c "import json;j=json.load(open('gpurun_out/r3k/r03_$i.json'));print('r03 code', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['wgrad']['frac'], j['roofline']['whole_step_frac'])"
done
