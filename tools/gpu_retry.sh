#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged)
# usage: tools/gpu_retry.sh <timeout seconds> <command string>
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
