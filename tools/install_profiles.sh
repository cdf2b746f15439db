#!/bin/bash
# Copies what tools/profile_r04.sh left under gpurun_out/prof_<tag>/ into profiles/<round>_* (the tracked summaries the README / DESIGN cite):
#   tools/install_profiles.sh <tag> [round, default r06]
# (the headers name the round the files are installed for; what earlier rounds measured is in their own profiles/rNN_* files)
set -eu
P=gpurun_out/prof_$1
R=${2:-r06}
RN=${R#r}; RN=$((10#$RN))
tail -1 $P/bench.json > profiles/${R}_bench.json
cp $P/bench_extra.json profiles/${R}_bench_extra.json 2>/dev/null || true
tail -2 $P/trace_bench.json > profiles/${R}_trace_bench.json
cp $P/trace/*/*_kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $P/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $P/headline_durations.txt profiles/${R}_headline_durations.txt
cp $P/scheme_pmc.json profiles/${R}_scheme_pmc.json
(echo "GPU suite WITHOUT the harness's two runtime variables (SC_TEST_NO_PIN_ENV=1 python3 -m pytest tests -m gpu -x -q; tools/profile_r04.sh), code of round $RN:"; tail -5 $P/gpu_suite_no_pin_env.txt) > profiles/${R}_gpu_suite_no_pin_env.txt
T() { python3 tools/timeline.py $P/tl_$1/trace/*/*kernel_trace.csv "$2" ${3:-}; }
(echo "Dilithium-III sign, batch 16384, one key: kernel timeline of one signing call, round $RN (rocprofv3 --kernel-trace over python3 tools/bench_scheme.py dil_sign 3, read with tools/timeline.py; one MI355X)."
 echo "Earlier: profiles/r03_dil_sign_timeline.txt 2707 us, profiles/r04_dil_sign_timeline.txt 2570 us."; echo; T dil_sign dil_mu_kernel) > profiles/${R}_dil_sign_timeline.txt
(echo "BLISS-B-I sign, batch 16384, one key, tapes resident: kernel timeline of one signing call, round $RN (rocprofv3 --kernel-trace over python3 tools/bench_scheme.py bliss_sign 3, read with tools/timeline.py; one MI355X)"
 echo "Earlier: profiles/r03_bliss_sign_timeline.txt, profiles/r04_bliss_sign_timeline.txt."; echo; T bliss_sign bl_zero_kernel 4096) > profiles/${R}_bliss_sign_timeline.txt
(echo "ring-TESLA-I sign, batch 16384, one key, tapes resident: kernel timeline of one signing call, round $RN (rocprofv3 --kernel-trace over python3 tools/bench_scheme.py tesla_sign 3, read with tools/timeline.py; one MI355X)"
 echo "Earlier: profiles/r03_tesla_sign_timeline.txt, profiles/r04_tesla_sign_timeline.txt."; echo; T tesla_sign closes:ts_zero6_kernel) > profiles/${R}_tesla_sign_timeline.txt
(echo "Falcon-1024 sign, batch 8192, one key, tapes resident: kernel timeline of one signing call, round $RN (rocprofv3 --kernel-trace over python3 tools/bench_scheme.py falcon_sign 3, read with tools/timeline.py; one MI355X)"; echo; T falcon_sign fal_hash2) > profiles/${R}_falcon_sign_timeline.txt
git diff --stat profiles | tail -3
