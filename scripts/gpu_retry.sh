#!/bin/bash
# usage: scripts/gpu_retry.sh TIMEOUT 'command'   -- retries only while gpurun reports "no slot free" (exit 3)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
