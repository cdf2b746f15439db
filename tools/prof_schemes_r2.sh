#!/bin/bash
# Per-kernel instruction counts and HBM traffic of the scheme-level legs (rocprofv3 --pmc, one counter group per pass, kernel trace in its own pass).
# Usage (GPU box, repository root): tools/prof_schemes_r2.sh <tag>  -> gpurun_out/prof_schemes_<tag>/ and a summary JSON
export TMPDIR=/tmp
OUT=gpurun_out/prof_schemes_$1
mkdir -p $OUT
LEGS="${LEGS:-dil_sign dil_verify bliss_sign falcon_sign polymul cdt}"
for leg in $LEGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$leg -- python3 tools/bench_scheme.py $leg 3 > $OUT/out_$leg.txt 2> $OUT/err_$leg.txt
  for c in SQ_INSTS_VALU SQ_WAVES FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${leg}_$c -- python3 tools/bench_scheme.py $leg 3 > $OUT/pmcout_${leg}_$c.txt 2> $OUT/pmcerr_${leg}_$c.txt
  done
done
python3 tools/summarize_scheme_pmc.py $OUT > $OUT/summary.json
cat $OUT/summary.json
