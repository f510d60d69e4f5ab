#!/bin/bash
# build the library as it was at a given commit into build/explib/libdetexhip_<tag>.so (same-run comparisons):
#   bash tools/build_lib_at.sh 0ebf32e r01      # the round-1 library
#   bash tools/build_lib_at.sh HEAD prev        # the last commit, against uncommitted changes
set -e
cd "$(dirname "$0")/.."
COMMIT=$1; TAG=$2; TMP=$(mktemp -d); mkdir -p build/explib
git archive "$COMMIT" detex_amd/csrc include | tar -x -C "$TMP"
if [ -f "$TMP/detex_amd/csrc/detexhip.hip" ]; then      # rounds 1-3: one translation unit
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -Wall -Wno-unused-function -Wno-pass-failed \
    -o build/explib/libdetexhip_$TAG.so "$TMP/detex_amd/csrc/detexhip.hip" "$TMP/detex_amd/csrc/ktx_loader.cpp"
else
  git archive "$COMMIT" Makefile | tar -x -C "$TMP"
  make -s -j8 -C "$TMP" lib LIB="$PWD/build/explib/libdetexhip_$TAG.so"
fi
rm -rf "$TMP"; ls -la build/explib/libdetexhip_$TAG.so
