#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# gpurun with retries while the pool is busy (exit code 3 / status transient: nothing charged).
# usage: tools/gpu_retry.sh <timeout-seconds> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
