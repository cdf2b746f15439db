#!/bin/bash
# Round 6, first GPU job: HEAD back under hardware evidence (rounds 4's end and all of 5 ran without a GPU).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/r06_gpu_first.sh'      ->  gpurun_out/r06a/
#  1. the whole GPU suite at HEAD    2. smoke()    3. the driver-shaped bench line    4. the opt-in finishing kernel's tests
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06a
mkdir -p $OUT
date > $OUT/started.txt
timeout 1200 python3 -m pytest tests -m gpu -q > $OUT/gpu_suite.txt 2>&1
echo "gpu suite rc=$?"; tail -6 $OUT/gpu_suite.txt
timeout 300 python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.out 2> $OUT/bench.err
echo "bench rc=$?"; tail -c 1500 $OUT/bench.out; cp bench_extra.json $OUT/ 2>/dev/null
SC_TEST_UNVALIDATED=1 timeout 600 python3 -m pytest tests/test_gpu_zz_dilithium_sparse_fin.py -q > $OUT/sparse_fin_tests.txt 2>&1
echo "sparse fin rc=$?"; tail -4 $OUT/sparse_fin_tests.txt
date > $OUT/finished.txt
