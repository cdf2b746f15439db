#!/bin/bash
# kernel-level rocprofv3 profile of one tool run:  tools/prof_one.sh <tag> <python script> [args...]
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/out.txt 2> $OUT/err.txt
tail -3 $OUT/out.txt
python3 - <<PY
import csv, glob
for f in glob.glob('$OUT/trace/*/*_kernel_stats.csv'):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), r['AverageNs'].rjust(14), r['Percentage'].rjust(7))
PY
