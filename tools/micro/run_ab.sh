#!/bin/bash
# A/B of alternative builds of libse_hip.so on ONE box: tools/micro/run_ab.sh <command...>
# runs the command with the tree's library, then with every tools/micro/bin/libse_hack*.so, then with the tree's again
cd "$(dirname "$0")/../.."
cp speech-enhancement_amd/libse_hip.so /tmp/libse_orig.so
for lib in /tmp/libse_orig.so tools/micro/bin/libse_hack*.so /tmp/libse_orig.so; do
  [ -f "$lib" ] || continue
  cp "$lib" speech-enhancement_amd/libse_hip.so
  touch speech-enhancement_amd/libse_hip.so
  echo "== $lib"
  "$@"
done
cp /tmp/libse_orig.so speech-enhancement_amd/libse_hip.so
