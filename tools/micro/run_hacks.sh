#!/bin/bash
# times `tools/microbench.py conv_one` with each experimental build of libse_hip.so found in tools/micro/bin/ (see conv3_ablate.py)
# usage: tools/micro/run_hacks.sh "<Cin> <precision> [planes]" ...   (library selected through SE_HIP_LIB, nothing is overwritten)
cd "$(dirname "$0")/../.."
for lib in "" tools/micro/bin/libse_hack*.so; do
  [ -z "$lib" ] || [ -f "$lib" ] || continue
  echo "== ${lib:-speech-enhancement_amd/libse_hip.so}"
  for c in "$@"; do
    if [ -z "$lib" ]; then python tools/microbench.py conv_one $c; else SE_HIP_LIB="$PWD/$lib" python tools/microbench.py conv_one $c; fi
  done
done
