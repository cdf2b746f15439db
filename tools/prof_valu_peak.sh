#!/bin/bash
# Settles the VALU issue ceiling with cycle counters:
#   1. tools/exp_valu_cycles alone: cycles per wave64 instruction per SIMD from s_memtime, the clock from s_memrealtime
#   2. the same binary (W = 1, 2, 4, 8 waves per SIMD) under rocprofv3 --pmc: what SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES /
#      SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE read on kernels whose issue rate is known from 1.
# Usage (GPU box, repository root): tools/prof_valu_peak.sh <tag>   -> gpurun_out/valu_peak_<tag>/
export TMPDIR=/tmp
OUT=gpurun_out/valu_peak_$1
mkdir -p $OUT
[ -x tools/exp_valu_cycles ] || hipcc -O2 --offload-arch=gfx950 -o tools/exp_valu_cycles tools/exp_valu_cycles.hip 2> /dev/null
rocprofv3 -L > $OUT/counters_list.txt 2>&1
tools/exp_valu_cycles > $OUT/cycles.txt 2> $OUT/cycles.err
cat $OUT/cycles.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc -- tools/exp_valu_cycles > $OUT/pmc_stdout.txt 2> $OUT/pmc_err.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- tools/exp_valu_cycles > $OUT/trace_stdout.txt 2> $OUT/trace_err.txt
python3 - "$OUT" <<'PY' | tee $OUT/pmc_summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in glob.glob(out + '/pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        key = (int(r['Dispatch_Id']), r['Kernel_Name'], int(r['Grid_Size']))
        rows.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
dur = {}
for f in glob.glob(out + '/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[int(r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
names = []
for line in open(out + '/cycles.txt'):                      # instruction i of the binary = the i-th distinct first word of its own report
    w = line.split()
    if w and w[0].startswith('v_') and w[0] not in names:
        names.append(w[0])
print('# per timed dispatch (iters 4096 x 32 instructions per wave): counters as rocprofv3 reports them (summed over the chip)')
print('# instr  W  waves  SQ_INSTS_VALU  SQ_ACTIVE_INST_VALU  ACTIVE/INSTS  SQ_BUSY_CYCLES  SQ_WAVE_CYCLES  GRBM_GUI_ACTIVE  INSTS_VALU/(GRBM/8)/1024 = wave-instr per SIMD per GRBM cycle | ACTIVE_INST_VALU*4/(GRBM/8*1024) | trace ns')
for (did, kn, grid), c in sorted(rows.items()):
    if c.get('SQ_INSTS_VALU', 0) < 1e6:      # the 64-trip warm-up launches
        continue
    import re
    m = re.search(r'k<(\d+)>', kn)
    op = names[int(m.group(1))] if m else kn[:20]
    W = grid // (256 * 256)
    g = c.get('GRBM_GUI_ACTIVE', 0) / 8.0
    iv, av = c.get('SQ_INSTS_VALU', 0), c.get('SQ_ACTIVE_INST_VALU', 0)
    print(f"{op:15s} {W} {c.get('SQ_WAVES',0):7.0f} {iv:14.0f} {av:14.0f} {av/iv if iv else 0:6.3f} {c.get('SQ_BUSY_CYCLES',0):14.0f} {c.get('SQ_WAVE_CYCLES',0):14.0f} {c.get('GRBM_GUI_ACTIVE',0):12.0f}"
          f"  {iv/g/1024 if g else 0:6.3f} | {av*4/(g*1024) if g else 0:6.3f} | {dur.get(did, 0)}")
PY
