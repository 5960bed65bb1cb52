#!/bin/bash
# usage: scratch/gpu_retry.sh TIMEOUT 'command'   -- retries while no GPU slot is free (gpurun exit code 3: nothing charged)
t=$1; shift
for i in $(seq 1 30); do
	/usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
	rc=$?
	if [ $rc -ne 3 ]; then exit $rc; fi
	sleep 90
done
exit 3
