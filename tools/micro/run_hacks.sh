#!/bin/bash
# times tools_conv_one.py with each experimental build of libse_hip.so found in tools/micro/bin/ (see conv3_ablate.py)
# usage: tools/micro/run_hacks.sh "<Cin> <precision> [planes]" ...
cd "$(dirname "$0")/../.."
cp speech-enhancement_amd/libse_hip.so /tmp/libse_orig.so
for lib in /tmp/libse_orig.so tools/micro/bin/libse_hack*.so; do
  cp "$lib" speech-enhancement_amd/libse_hip.so
  echo "== $lib"
  for c in "$@"; do python tools/tools_conv_one.py $c; done
done
cp /tmp/libse_orig.so speech-enhancement_amd/libse_hip.so
