export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -k "conv" 2>&1 | grep -E "passed|failed|^E  |^FAILED|Error" | tail -8
for v in "" xburst "" xburst; do
  echo "variant [$v]"; if [ -z "$v" ]; then timeout 200 python scratch/layer_time.py | cut -c1-60; else CONVASR_HIP_LIB=$PWD/convasr_amd/libconvasr_hip.$v.so timeout 200 python scratch/layer_time.py | cut -c1-60; fi
done
