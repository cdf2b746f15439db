#!/bin/bash
# Round 5, first GPU job (the pool was closed to this build from the last hours of round 4 on; tools/poll_gpu.sh launches this the moment it opens):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r05_gpu_queue.sh'      ->  gpurun_out/r05q/
#  1. the whole GPU suite at HEAD (the default Dilithium path has not run on hardware since 111b673)
#  2. tools/validate_sparse_fin.sh: the table finishing kernel's opt-in tests, A/B of the north-star leg, kernel timeline  -> gpurun_out/sparse_fin/
#  3. the driver-shaped bench line
#  4. tools/profile_r04.sh r05: kernel trace, PMC traffic, per-leg counters, timelines                                   -> gpurun_out/prof_r05/
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r05q
mkdir -p $OUT
date > $OUT/started.txt
timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_suite.txt 2>&1
echo "gpu suite rc=$?"; tail -4 $OUT/gpu_suite.txt
timeout 1500 bash tools/validate_sparse_fin.sh 2>&1 | tail -30
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.out 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 1200 $OUT/bench.out; cp bench_extra.json $OUT/ 2>/dev/null
timeout 2400 bash tools/profile_r04.sh r05 2>&1 | tail -25
date > $OUT/finished.txt
