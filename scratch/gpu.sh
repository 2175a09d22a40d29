#!/bin/bash
# scratch/gpu.sh TIMEOUT SCRIPT  -- gpurun with retries while no GPU slot is free (rc 3)
T=$1; S=$2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "bash $S" > /tmp/gpu_last.log 2>&1
  rc=$?
  if [ $rc -ne 3 ] && ! grep -q "status=transient" /tmp/gpu_last.log; then break; fi
  sleep 90
done
tail -150 /tmp/gpu_last.log
echo "gpurun rc=$rc after $i attempt(s)"
