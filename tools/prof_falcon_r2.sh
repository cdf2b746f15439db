#!/bin/bash
# Falcon-1024, batch 8192 (BASELINE configs[4]): kernel trace + HBM traffic (separate PMC passes) of the sampler, for the default mapping
# (8 signatures per wavefront, lower tree in LDS: lanes = 64) and for the first-generation 8-lanes-per-signature mapping (lanes = 8).
# Usage: tools/prof_falcon_r2.sh <tag>     (run from the repository root on the GPU box; writes gpurun_out/prof_falcon_<tag>/)
export TMPDIR=/tmp
OUT=gpurun_out/prof_falcon_$1
mkdir -p $OUT
for L in 64 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$L -- python3 tools/bench_falcon.py cfg4 $L > $OUT/out_$L.txt 2> $OUT/err_$L.txt
  grep sign_ops $OUT/out_$L.txt
  cut -c1-200 $OUT/trace_$L/*/*_kernel_stats.csv | head -8
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${L}_$c -- python3 tools/bench_falcon.py cfg4 $L > /dev/null 2> $OUT/err_${L}_$c.txt
    python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('$OUT/pmc_${L}_$c/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:4]:
    print('lanes=$L $c', k, 'dispatches', n, 'per_dispatch', v / n)
PY
  done
done
