#!/bin/bash
# Round-2 profile evidence (run on the GPU box through gpurun; kernel trace / stats and PMC counters in separate runs, as the pool requires):
#   tools/profile_r02.sh <tag>   ->  gpurun_out/prof_<tag>/
#  1. the driver-shaped bench line                                   bench.json
#  2. rocprofv3 --kernel-trace --stats of the same command           trace/  (no CPU baseline children under the profiler)
#  3. FETCH_SIZE / WRITE_SIZE of the headline kernel                 pmc_fetch/ pmc_write/
#  4. per-kernel instruction counts and traffic of the scheme legs   (tools/prof_schemes_r2.sh)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_write.err
python3 tools/summarize_headline_pmc.py $OUT > $OUT/pmc_traffic.json
cat $OUT/pmc_traffic.json
cut -c1-170 $OUT/trace/*/*_kernel_stats.csv | head -12
