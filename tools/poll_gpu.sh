#!/bin/bash
# Retries a GPU job until the pool accepts it (a refused call costs nothing): tools/poll_gpu.sh [seconds between attempts] [script] [gpurun timeout]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
JOB=${2:-tools/r06_gpu_first.sh}
while true; do
  /usr/local/graft/bin/gpurun --timeout ${3:-2700} -- "bash $JOB" > gpurun_out/poll_call.log 2>&1
  rc=$?
  echo "$(date) $JOB rc=$rc" >> gpurun_out/poll.log
  if [ $rc -ne 2 ] && [ $rc -ne 3 ]; then exit $rc; fi
  sleep ${1:-600}
done
