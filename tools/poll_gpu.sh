#!/bin/bash
# Retries the round's first GPU job until the pool accepts it (a refused call costs nothing): tools/poll_gpu.sh [seconds between attempts]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
while true; do
  /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r05_gpu_queue.sh' > gpurun_out/r05q_call.log 2>&1
  rc=$?
  echo "$(date) rc=$rc" >> gpurun_out/poll.log
  if [ $rc -ne 2 ] && [ $rc -ne 3 ]; then exit $rc; fi
  sleep ${1:-900}
done
