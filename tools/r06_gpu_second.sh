#!/bin/bash
# Round 6, second GPU job (after tools/r06_gpu_first.sh has shown the suite green at HEAD):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/r06_gpu_second.sh'      ->  gpurun_out/sparse_fin/, gpurun_out/prof_r06/, gpurun_out/r06b/
#  1. tools/validate_sparse_fin.sh: A/B of the north-star leg with the finishing kernel without transforms (tuning bit 11), its timeline
#  2. the tuning experiment of DESIGN 8 (2 / 6 / 32 slots per message against the default schedule)
#  3. tools/profile_r04.sh r06: kernel trace, PMC traffic, per-leg counters, timelines -> install with tools/install_profiles.sh r06 r06
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06b
mkdir -p $OUT
timeout 1500 bash tools/validate_sparse_fin.sh 2>&1 | tail -30
for t in 0 2 3; do SC_BENCH_DIL_TUNING=$t timeout 200 python3 tools/bench_scheme.py dil_sign 7 > $OUT/slots_$t.txt 2>&1; echo "slots $t: $(tail -1 $OUT/slots_$t.txt)"; done
timeout 2400 bash tools/profile_r04.sh r06 2>&1 | tail -25
date > $OUT/finished.txt
