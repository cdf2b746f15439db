#!/bin/bash
# SQ counters of the kernels whose name contains <substring>, one rocprofv3 --pmc pass per group:
#   tools/prof_kernel_pmc.sh <tag> <substring> <program> [args...]     (the program after -- must be the real one: python3 script.py ...)
export TMPDIR=/tmp
TAG=$1; shift
KSUB=$1; shift
OUT=gpurun_out/kpmc_$TAG
mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- "$@" > $OUT/out_g$i.txt 2> $OUT/err_g$i.txt
done
python3 - "$OUT" "$KSUB" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + '/g*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if ksub not in r['Kernel_Name']: continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:40]
        a = acc[k][r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, d in acc.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print(f'   {c:24s} dispatches {n:4d}  per dispatch {v / n:16.0f}')
    g = lambda c: d[c][1] / max(d[c][0], 1) if c in d else 0.0
    if g('SQ_WAVE_CYCLES'):
        wc = g('SQ_WAVE_CYCLES')
        print(f"   of a wavefront's cycles: waiting (s_waitcnt / barrier) {g('SQ_WAIT_ANY') / wc:.2f}, issue stalls {g('SQ_WAIT_INST_ANY') / wc:.2f}, issuing {g('SQ_ACTIVE_INST_ANY') / wc:.2f}")
        print(f"   VALU instructions per SIMD-cycle (SQ_INSTS_VALU / (SQ_BUSY_CYCLES / 32 x 1024)): {g('SQ_INSTS_VALU') / (g('SQ_BUSY_CYCLES') / 32 * 1024):.3f}   (ceilings measured by tools/exp_valu_cycles: 0.40-0.43 full-rate class, 0.235 the rest)")
PY
