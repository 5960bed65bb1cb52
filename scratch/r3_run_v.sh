for i in 1 2; do
X_SKIP=0 timeout 200 python scratch/ab_nofinalize.py 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with finalize', j['ms_per_step'])"
X_SKIP=1 timeout 200 python scratch/ab_nofinalize.py 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip finalize', j['ms_per_step'])"
done
