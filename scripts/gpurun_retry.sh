#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3 = nothing charged): scripts/gpurun_retry.sh <timeout> '<command>'
set -o pipefail
T=${1:?timeout}; shift
for k in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
