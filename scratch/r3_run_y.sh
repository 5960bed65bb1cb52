for i in 1 2 3; do
for m in 0 1; do CONVASR_OVERLAP_REDUCE=$m timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-traffic --no-kernel-timer 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $m', j['ms_per_step'], j.get('loss'))"; done
done
CONVASR_OVERLAP_REDUCE=1 timeout 900 python -m pytest tests/test_models_gpu.py tests/test_round2_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|^E  |^FAILED|Error" | tail -5
