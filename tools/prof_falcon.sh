#!/bin/bash
# kernel-level profile of the Falcon path (tools/bench_falcon.py), then FETCH/WRITE traffic in separate PMC passes
export TMPDIR=/tmp
OUT=gpurun_out/prof_falcon_$1
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_falcon.py > $OUT/out.txt 2> $OUT/err.txt
cat $OUT/out.txt
cut -c1-170 $OUT/trace/*/*_kernel_stats.csv | head -12
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 tools/bench_falcon.py one > /dev/null 2> $OUT/err_$c.txt
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob('$OUT/pmc_$c/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:6]:
    print('$c', k, 'dispatches', n, 'sum', v, 'per_dispatch', v / n)
PY
done
