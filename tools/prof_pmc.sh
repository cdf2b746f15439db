#!/bin/bash
# PMC counters (one pass per group, each under its own timeout) of one tool run:
#   tools/prof_pmc.sh <tag> <kernel-substring> <python script> [args...]
export TMPDIR=/tmp
TAG=$1; shift
KSUB=$1; shift
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_')
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/$name -- python3 "$@" > /dev/null 2> $OUT/err_$name.txt
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob('$OUT/$name/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if '$KSUB' not in r['Kernel_Name']: continue
        k = r['Kernel_Name'][:48]
        a = acc[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in acc.items():
    for c, (n, v) in d.items():
        print(k, c, 'dispatches', n, 'per_dispatch', v / n)
PY
done
