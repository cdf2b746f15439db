#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/prof_dil_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_dilithium.py > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
cut -c1-150 $OUT/trace/*/*_kernel_stats.csv | head -30
