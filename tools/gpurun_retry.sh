#!/bin/bash
# tools/gpurun_retry.sh LOG TIMEOUT 'command': gpurun with retries while every GPU slot of the pod is busy (exit code 3)
log=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
