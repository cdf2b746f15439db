#!/bin/bash
# First GPU job of the next round (run on the GPU box through gpurun): the Dilithium finishing kernel without transforms (csrc/dil_cfins.h, tuning bit 11)
# has only run on the CPU wave emulation so far.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/validate_sparse_fin.sh'   ->  gpurun_out/sparse_fin/
#  1. its opt-in GPU tests (reference signatures under all four parameter sets; byte equality with the transform kernel at batch 1 / 257 / 5000; a key per message)
#  2. A/B of the north-star leg (and of the distinct-keys leg): alternating pairs of runs on the same box (medians inside bench_scheme.py)
#  3. kernel timeline of one signing call with the bit set
# If 1 is green and 2 says faster: make `sparse_fin` the one-key default in sign_core (csrc/dilithium.hip), drop the skip in
# tests/test_gpu_zz_dilithium_sparse_fin.py, re-run the GPU suite, tools/profile_r04.sh and tools/install_profiles.sh, refresh README / DESIGN 4.4.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/sparse_fin
mkdir -p $OUT
SC_TEST_UNVALIDATED=1 timeout 600 python3 -m pytest tests/test_gpu_zz_dilithium_sparse_fin.py -x -q > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
for i in 1 2 3; do
  timeout 200 python3 tools/bench_scheme.py dil_sign 9 > $OUT/ntt_fin_$i.txt 2>&1
  SC_BENCH_DIL_TUNING=2048 timeout 200 python3 tools/bench_scheme.py dil_sign 9 > $OUT/sparse_fin_$i.txt 2>&1
  tail -1 $OUT/ntt_fin_$i.txt; tail -1 $OUT/sparse_fin_$i.txt
done
for i in 1 2; do
  timeout 300 python3 tools/bench_scheme.py dil_sign_keys 7 > $OUT/keys_ntt_fin_$i.txt 2>&1
  SC_BENCH_DIL_TUNING=2048 timeout 300 python3 tools/bench_scheme.py dil_sign_keys 7 > $OUT/keys_sparse_fin_$i.txt 2>&1
  tail -1 $OUT/keys_ntt_fin_$i.txt; tail -1 $OUT/keys_sparse_fin_$i.txt
done
mkdir -p $OUT/tl
export SC_BENCH_DIL_TUNING=2048
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl/trace -- python3 tools/bench_scheme.py dil_sign 3 > $OUT/tl/out.txt 2> $OUT/tl/err.txt
python3 tools/timeline.py $OUT/tl/trace/*/*kernel_trace.csv dil_mu_kernel 16384 > $OUT/timeline.txt 2>&1; cat $OUT/timeline.txt
