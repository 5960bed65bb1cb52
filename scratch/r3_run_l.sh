python scratch/layer_time_wgrad_all.py 2>&1 | tail -16
