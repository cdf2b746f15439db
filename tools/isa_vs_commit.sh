#!/bin/bash
# Is the device code of the working tree the device code of <commit>?  tools/isa_vs_commit.sh <commit>
# Every .hip of csrc/Makefile is compiled device-only to a gfx950 listing in both trees (the commit's from `git archive`) and compared part by part
# (tools/isa_cmp.py: function bodies, kernel descriptors, data objects, the metadata note's kernel entries; labels, comments and the compilation unit's id
# normalised).  Used in rounds 5 and 6 - rounds without a GPU - to show that the kernels at HEAD are the kernels of the last commit the GPU suite ran on,
# whatever headers their text has moved to since.  A compile that fails is reported (its stderr is kept and shown) and fails the run.
set -u
cd "$(dirname "$0")/.."
C=${1:?commit}
OLD=$(mktemp -d); A=$(mktemp -d); B=$(mktemp -d)
git archive "$C" libsafecrypto_amd/csrc include | tar -x -C "$OLD"
FILES=$(sed -n 's/^SRC *:= *//p' libsafecrypto_amd/csrc/Makefile | tr " " "\n" | sed -n 's/\.hip$//p' | tr "\n" " ")       # every .hip the product is built from now
for f in $FILES; do
  fl=""; [ $f = falcon ] && fl="-ffp-contract=off"
  [ -e "$OLD/libsafecrypto_amd/csrc/$f.hip" ] && (cd "$OLD/libsafecrypto_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $fl --cuda-device-only -S $f.hip -o "$A/$f.s" 2> "$A/$f.err" || echo "compile failed" >> "$A/$f.err.rc") &
  (cd libsafecrypto_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $fl --cuda-device-only -S $f.hip -o "$B/$f.s" 2> "$B/$f.err" || echo "compile failed" >> "$B/$f.err.rc") &
  while (( $(jobs -r | wc -l) >= ${JOBS:-6} )); do wait -n; done
done
wait
echo "device code: $C (first) against the working tree at $(git rev-parse --short HEAD)$(git diff --quiet -- libsafecrypto_amd include || echo ' + uncommitted changes') (second)"
rc=0
for f in $FILES; do
  echo -n "$f.hip: "
  if [ ! -e "$OLD/libsafecrypto_amd/csrc/$f.hip" ]; then
    if [ -e "$B/$f.err.rc" ] || [ ! -s "$B/$f.s" ]; then echo "COMPILE FAILED:"; grep -v "argument unused during compilation" "$B/$f.err" | head -20; rc=1
    else echo "a file $C does not have: $(grep -c '^[[:space:]]*\.amdhsa_kernel ' "$B/$f.s") kernels, all additions"; fi
    continue
  fi
  for side in "$A" "$B"; do
    if [ -e "$side/$f.err.rc" ] || [ ! -s "$side/$f.s" ]; then echo "COMPILE FAILED ($side):"; grep -v "argument unused during compilation" "$side/$f.err" | head -20; rc=1; fi
  done
  python3 tools/isa_cmp.py --allow-additions "$A/$f.s" "$B/$f.s" || rc=1
done
rm -rf "$OLD" "$A" "$B"
[ $rc -eq 0 ] && echo "RESULT: every function body, kernel descriptor, data object and metadata entry of $C is unchanged in the working tree (additions listed above, if any)" || echo "RESULT: differences or failures above"
exit $rc
