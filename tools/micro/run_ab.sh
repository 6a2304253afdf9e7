#!/bin/bash
# A/B of alternative builds of libse_hip.so on ONE box: tools/micro/run_ab.sh <command...>
# runs the command with the tree's library, then with every tools/micro/bin/libse_hack*.so (selected through SE_HIP_LIB,
# read by speech-enhancement_amd/_lib.py -- the product library is never overwritten), then with the tree's again
cd "$(dirname "$0")/../.."
for lib in "" tools/micro/bin/libse_hack*.so ""; do
  [ -z "$lib" ] || [ -f "$lib" ] || continue
  echo "== ${lib:-speech-enhancement_amd/libse_hip.so}"
  if [ -z "$lib" ]; then "$@"; else SE_HIP_LIB="$PWD/$lib" "$@"; fi
done
