ls /sys/class/drm/ | head; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_*; do echo $f; cat $f 2>&1 | head -2; done
rocm-smi --showpower --showclocks 2>&1 | head -30
(python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-traffic --no-kernel-timer > gpurun_out/pw_bench.json 2>/dev/null &) ; sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -6; for f in /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input; do [ -r $f ] && echo "$f $(cat $f)"; done; sleep 0.5; done
wait; sleep 3; cut -c1-200 gpurun_out/pw_bench.json
