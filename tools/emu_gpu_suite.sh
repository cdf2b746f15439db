#!/bin/bash
# The whole `-m gpu` suite against the product library built for the CPU (tests/emu_harness.py build_library: host code + kernels over tests/emu/hip_host_emu.h,
# a host thread per lane; tests/conftest.py SC_EMU_LIB points the binding at it and torch's "cuda" at the CPU).  Test infrastructure - about an hour on 8 cores.
#   tools/emu_gpu_suite.sh [plain|asan|late]      ->  gpurun_out/emu_suite/<kind>.txt ; copy the summary under profiles/ to keep it
# late: the plain build with EMU_SIDE_STREAMS=late - work on the library's own streams held back as long as the HIP model allows (tests/test_emu_stream_order.py)
# Deselected: what only means something on the device (the bench line, a C host linked against the device library, the glue over the reference's own build,
# graph capture, two ranks sharing a device) - and listed at the end of the output: the tests that did not finish inside the per-test limit (the full
# BASELINE batches: a 16384-message signing call is days of emulation).
set -u
cd "$(dirname "$0")/.."
KIND=${1:-plain}
SAN=""; [ "$KIND" = asan ] && SAN="address,undefined"
LIB=$(python3 -c "
import sys; sys.path.insert(0, '.')
from tests import emu_harness as H
print(H.build_library('$SAN' or None))" | tail -1)
mkdir -p gpurun_out/emu_suite
ENVV="SC_EMU_LIB=$LIB"
[ "$KIND" = late ] && ENVV="$ENVV EMU_SIDE_STREAMS=late"
if [ -n "$SAN" ]; then
  CL=/opt/rocm/lib/llvm/bin/clang++
  ENVV="$ENVV LD_PRELOAD=$($CL -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1"
fi
env $ENVV python3 -m pytest tests -m gpu -q -n ${JOBS:-7} --timeout ${LIMIT:-600} --timeout-method=thread -p no:cacheprovider \
  --deselect tests/test_gpu_multi.py::test_bench_two_ranks_contract_line --deselect tests/test_gpu_00_two_ranks.py --deselect tests/test_gpu_c_host.py \
  --deselect tests/test_gpu_dropin.py --deselect tests/test_gpu_ntt.py::test_graph_replayed_steps_equal_single_call \
  > gpurun_out/emu_suite/$KIND.txt 2>&1
grep -a "^FAILED\| passed\| failed" gpurun_out/emu_suite/$KIND.txt | sed 's/ - .*//' | tail -60
