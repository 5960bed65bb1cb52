export PYTHONPATH=$PWD
for v in "" xnodma ""; do
  echo "variant [$v]"; if [ -z "$v" ]; then timeout 300 python scratch/layer_time_wgrad_all.py | cut -c1-70; else CONVASR_HIP_LIB=$PWD/convasr_amd/libconvasr_hip.$v.so timeout 300 python scratch/layer_time_wgrad_all.py | cut -c1-70; fi
done
