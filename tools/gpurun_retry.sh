#!/bin/bash
# Retry `gpurun` while the pod has no free GPU slot (exit code 3: nothing ran, nothing charged).  Never retries a call that ran.
# usage: tools/gpurun_retry.sh <timeout_s> '<command>'
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 150
done
exit 3
