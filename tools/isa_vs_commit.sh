#!/bin/bash
# Is the device code of the working tree the device code of <commit>?  tools/isa_vs_commit.sh <commit>
# Every .hip of csrc/Makefile is compiled device-only to a gfx950 listing in both trees (the commit's from `git archive`) and compared function by function
# (tools/isa_cmp.py: labels, comments and the compilation unit's id normalised).  Used in round 5 - a round without a GPU - to show that the kernels at HEAD are the
# kernels of the last commit the GPU suite ran on, whatever headers their text has moved to since.
set -u
cd "$(dirname "$0")/.."
C=${1:?commit}
OLD=$(mktemp -d); A=$(mktemp -d); B=$(mktemp -d)
git archive "$C" libsafecrypto_amd/csrc include | tar -x -C "$OLD"
for f in sc_hip dilithium kyber bliss falcon ops drbg tesla rlwe ens dlp entropy; do
  fl=""; [ $f = falcon ] && fl="-ffp-contract=off"
  (cd "$OLD/libsafecrypto_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $fl --cuda-device-only -S $f.hip -o "$A/$f.s" 2>/dev/null) &
  (cd libsafecrypto_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $fl --cuda-device-only -S $f.hip -o "$B/$f.s" 2>/dev/null) &
  while (( $(jobs -r | wc -l) >= 6 )); do wait -n; done
done
wait
echo "device code: $C (first) against the working tree at $(git rev-parse --short HEAD) (second)"
for f in sc_hip dilithium kyber bliss falcon ops drbg tesla rlwe ens dlp entropy; do echo -n "$f.hip: "; python3 tools/isa_cmp.py "$A/$f.s" "$B/$f.s"; done
rm -rf "$OLD" "$A" "$B"
