#!/bin/bash
# Build the *reference itself* (libsafecrypto, CPU C99) into oracle/_ref/ so that the
# oracle restatement in this directory and the golden fixtures under tests/golden/ can
# be pinned against it.  TEST INFRASTRUCTURE ONLY: nothing in the product path links it.
#
# The reference sources are compiled where they lie under $REF (read-only); every
# intermediate (pre-processed NTT template instances, the table generator's output,
# objects) goes to a scratch directory outside the repo, and only binaries
# (libsafecrypto_ref.so, ref_driver) land in oracle/_ref/ (git-ignored).
# No reference source is copied into the repo, and no stand-in for any header,
# library or tool is written: the only generated inputs are produced by the
# reference's own generators compiled from its own sources
#   - src/utils/arith/ntt_template.{c,h}.in expanded by the C preprocessor with the
#     -D switches listed in src/utils/arith/gen_ntt.sh:41-103 (what autogen.sh:4 does)
#   - build_tools/ntt_table_gen/main.c  -> ntt_tables.{c,h}
#   - setup_version.sh.in               -> safecrypto_version.h
# Flags follow SURVEY.md section 8(c).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${REF:-/root/reference}"
OUT="$HERE/_ref"
B="${SC_REF_BUILD:-/tmp/sc_ref_build}"
if [ ! -d "$REF/src" ]; then
    echo "build_ref: $REF not present - keeping prebuilt oracle/_ref (if any)"; exit 0
fi
rm -rf "$B"; mkdir -p "$B/gen/utils/arith" "$OUT"
CC="${CC:-gcc}"
DEFS="-D_GNU_SOURCE -DHAVE_64BIT -DHAVE_128BIT -DUSE_SAFECRYPTO_INTEGER_MP -DUSE_SAFECRYPTO_FLOAT_MP \
 -DSHA3_UNROLLED -DHAVE___BUILTIN_CTZ -DHAVE___BUILTIN_CLZ -DCONFIGURE_INVOCATION=\"manual\" -DECRYPT_API=\"ecrypt-sync.h\" \
 -DNTT_NEEDS_16813057 -DNTT_NEEDS_134348801"   # the last two: keep the tables of the DLP-IBE expanded sets 6 / 7 in the library (fixtures only)
INC="-I$B/gen -I$B/gen/utils/arith -I$REF/src/utils/arith -I$REF/include -I$REF/src -I$REF/src/utils/crypto -I$REF"
CFLAGS="-O2 -std=gnu11 -fcommon -fPIC -w $DEFS $INC"

# 1. template instances (same switches as gen_ntt.sh; output outside the repo)
A="$REF/src/utils/arith"
gen() { # name switch
    local n=$1 sw=$2
    {
      printf '#include "utils/arith/ntt.h"\n#ifndef USE_RUNTIME_NTT_TABLES\n#include "utils/arith/ntt_tables.h"\n#endif\n'
      printf '#include "utils/arith/sc_math.h"\n#include <stdio.h>\n#include "utils/arith/ntt_%s.h"\n' "$n"
      cpp -CC -P -nostdinc -E "$A/ntt_template.c.in" -DNTT_NAME_EXT="$n" "$sw" -D_NTT_MK1
    } > "$B/gen/utils/arith/ntt_$n.c"
    {
      printf '#pragma once\n#include "safecrypto_types.h"\n#include "utils/arith/limb.h"\n'
      cpp -CC -P -nostdinc -E "$A/ntt_template.h.in" -DNTT_NAME_EXT="$n" "$sw" -D_NTT_MK1
    } > "$B/gen/utils/arith/ntt_$n.h"
}
gen reference -D_ENABLE_REFERENCE
gen barrett   -D_ENABLE_BARRETT
gen fp        -D_ENABLE_FLOATING_POINT
gen 7681      -D_ENABLE_SOLINAS_7681
gen 8380417   -D_ENABLE_SOLINAS_8380417

# 2. version header by the reference's own script
bash "$REF/setup_version.sh.in" "$REF/version" "$B/gen/safecrypto_version.h"

# 3. NTT tables by the reference's own generator
$CC -O2 -w $DEFS $INC "$REF/build_tools/ntt_table_gen/main.c" \
    "$A/roots_of_unity.c" "$A/limb.c" "$A/limb_base.c" "$A/sc_math.c" -lm -o "$B/ntt_table_gen"
"$B/ntt_table_gen" "$B/gen/utils/arith/ntt_tables" > /dev/null

# 4. every library source named by the reference's Makefile.am files
# (token scan of the un-commented lines of each Makefile.am under src/, unit tests excluded;
#  generated files are compiled from the scratch dir below instead)
( cd "$REF/src" && for m in $(find . -name Makefile.am | grep -v /unit); do d=$(dirname "$m");
    grep -v '^[[:space:]]*#' "$m" | { grep -oE '[A-Za-z0-9_/.-]+\.c\b' || true; } | while read -r f; do
      if [ -f "$d/$f" ]; then echo "$d/$f"; fi; done; done | sort -u | grep -v -e ntt_avx -e isaac/rand.c ) > "$B/srcs.txt"
compile_one() {
    local s=$1 o
    o="$OBJ/$(echo "${s#./}" | tr '/' '_' | sed 's/\.c$/.o/')"
    $CC $CFLAGS -c "$REF/src/$s" -o "$o" 2> "$o.err" || { echo "FAIL $s"; cat "$o.err"; return 1; }
}
export -f compile_one
# one library + driver per build flavour: the plain build (what the fixtures are generated with) and the AVX2 build of SURVEY 8(d)'s CPU protocol
build_flavour() { # suffix extra-flags
    local sfx=$1 extra=$2
    export OBJ="$B/obj$sfx"; mkdir -p "$OBJ"
    export CFLAGS="-O2 -std=gnu11 -fcommon -fPIC -w $DEFS $extra $INC"
    export CC REF B
    xargs -P "$(nproc)" -I{} bash -c 'compile_one {}' < "$B/srcs.txt"
    for f in "$B"/gen/utils/arith/ntt_*.c; do
        case "$f" in *ntt_avx.c) [ -n "$sfx" ] || continue;; esac
        $CC $CFLAGS -c "$f" -o "$OBJ/gen_$(basename "${f%.c}").o"
    done
    $CC -shared -o "$OUT/libsafecrypto_ref$sfx.so" "$OBJ"/*.o -lm -lpthread
    # 5. our driver (our source, reference headers at build time only)
    $CC -O2 -std=gnu11 -w $DEFS $extra $INC "$HERE/ref_driver.c" -o "$OUT/ref_driver$sfx" \
        -L"$OUT" -lsafecrypto_ref$sfx -Wl,-rpath,'$ORIGIN' -lm -lpthread
}
build_flavour "" ""
if grep -q avx2 /proc/cpuinfo; then
    {   # the AVX2 instance of the template, as gen_ntt.sh:95-103 makes it
      printf '#include "utils/arith/ntt.h"\n#ifndef USE_RUNTIME_NTT_TABLES\n#include "utils/arith/ntt_tables.h"\n#endif\n'
      printf '#include "utils/arith/sc_math.h"\n#include <stdio.h>\n#include "utils/arith/ntt_avx.h"\n#include <immintrin.h>\n'
      cpp -CC -P -nostdinc -E "$A/ntt_template.c.in" -DNTT_NAME_EXT="avx" -D_ENABLE_AVX2_INTRINSICS -D_NTT_MK1
    } > "$B/gen/utils/arith/ntt_avx.c"
    {
      printf '#pragma once\n#include "safecrypto_types.h"\n#include "utils/arith/limb.h"\n'
      cpp -CC -P -nostdinc -E "$A/ntt_template.h.in" -DNTT_NAME_EXT="avx" -D_ENABLE_AVX2_INTRINSICS -D_NTT_MK1
    } > "$B/gen/utils/arith/ntt_avx.h"
    build_flavour "_avx2" "-mavx2 -DHAVE_AVX2 -DHAVE_AVX" || echo "build_ref: AVX2 flavour failed - the plain build stands"
fi
# 6. the drop-in proof (examples/safecrypto_glue: the libsafecrypto-side binding + a driver; our sources, reference headers at build time only),
#    linked with the reference library AND the engine; it only runs where there is a GPU (tests/test_gpu_dropin.py)
ENG="$HERE/../libsafecrypto_amd/lib"
GLUE="$HERE/../examples/safecrypto_glue"
if [ -f "$ENG/libsc_hip.so" ]; then
    $CC -O2 -std=gnu11 -w $DEFS $INC -I"$HERE/../include" -I"$GLUE" "$GLUE/dropin_driver.c" "$GLUE/sc_hip_glue.c" -o "$OUT/dropin_driver" \
        -L"$OUT" -lsafecrypto_ref -L"$ENG" -lsc_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,'$ORIGIN/../../libsafecrypto_amd/lib' \
        -Wl,-rpath-link,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lm -lpthread
else
    echo "build_ref: libsc_hip.so not built yet - skipping dropin_driver"
fi
[ -n "${SC_REF_KEEP:-}" ] || rm -rf "$B"
echo "build_ref: OK -> $OUT"
