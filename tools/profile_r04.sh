#!/bin/bash
# Round-4 profile evidence (run on the GPU box through gpurun; kernel trace / stats and PMC counters in separate runs, as the pool requires):
#   tools/profile_r04.sh <tag>   ->  gpurun_out/prof_<tag>/
#  1. the driver-shaped bench line (its scheme rooflines profile their own legs: tools/pmc_leg.py children)   bench.json
#  2. rocprofv3 --kernel-trace --stats of the same command (no CPU baseline / counter children under the profiler)  trace/
#  3. FETCH_SIZE / WRITE_SIZE of the headline kernel                                                          pmc_fetch/ pmc_write/ -> pmc_traffic.json
#  4. the per-leg counter passes as a file of their own (bench.py's fallback when rocprofv3 is absent)         scheme_pmc.json
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_$1
mkdir -p $OUT
timeout 900 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cp bench_extra.json $OUT/bench_extra.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $OUT/pmc_write.err
python3 tools/summarize_headline_pmc.py $OUT durations > $OUT/pmc_traffic.json 2> $OUT/headline_durations.txt
python3 - "$OUT" <<'PY'
import json, sys
sys.path.insert(0, 'tools')
import pmc_leg
out = {}
for leg in ('dil_sign', 'dil_verify', 'bliss_sign', 'bliss_verify', 'falcon_sign', 'ens_sign', 'rlwe_encrypt', 'polymul', 'drbg', 'tesla_sign', 'tesla_verify'):
    r = pmc_leg.profile_leg(leg, calls=2)
    if r.get('ok'):
        r.pop('ok')
        out[leg] = r
json.dump(out, open(sys.argv[1] + '/scheme_pmc.json', 'w'), indent=1)
print({k: (round(v['valu_wave_instructions_per_op']), round(v['hbm_bytes_per_op_2xfetch_plus_write'])) for k, v in out.items()})
PY
# kernel timelines of one call of the north-star legs, and the GPU suite once WITHOUT the harness's two runtime variables
for leg in dil_sign bliss_sign falcon_sign tesla_sign; do
  mkdir -p $OUT/tl_$leg
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$leg/trace -- python3 tools/bench_scheme.py $leg 3 > $OUT/tl_$leg/out.txt 2> $OUT/tl_$leg/err.txt
done
SC_TEST_NO_PIN_ENV=1 timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/gpu_suite_no_pin_env.txt 2>&1
tail -3 $OUT/gpu_suite_no_pin_env.txt
cat $OUT/pmc_traffic.json; cat $OUT/headline_durations.txt
cut -c1-170 $OUT/trace/*/*_kernel_stats.csv | head -14
