#!/bin/bash
# rocprofv3 passes for the round's profile evidence (run on the GPU box through gpurun).
# kernel-trace/stats and PMC counters are collected in separate runs, as the pool requires.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
hipcc --offload-arch=gfx950 -O3 -o /tmp/exp_io tools/exp_io.hip 2>/dev/null
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/cal_fetch -- /tmp/exp_io > $OUT/cal_exp_io.txt 2> $OUT/cal_fetch.err
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/cal_write -- /tmp/exp_io > /dev/null 2> $OUT/cal_write.err
find $OUT -name "*.csv" | head -40
du -sh $OUT
