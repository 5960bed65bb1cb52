export PYTHONPATH=$PWD
for v in "" xrev "" xrev; do
  echo "variant [$v]"; if [ -z "$v" ]; then timeout 200 python scratch/layer_time.py | cut -c1-60; else CONVASR_HIP_LIB=$PWD/convasr_amd/libconvasr_hip.$v.so timeout 200 python scratch/layer_time.py | cut -c1-60; fi
done
