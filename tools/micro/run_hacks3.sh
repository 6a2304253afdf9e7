#!/bin/bash
cd "$(dirname "$0")/../.."
cp speech-enhancement_amd/libse_hip.so /tmp/libse_orig.so
for lib in /tmp/libse_orig.so tools/micro/bin/libse_hack*.so /tmp/libse_orig.so; do
  cp "$lib" speech-enhancement_amd/libse_hip.so
  echo "== $lib"
  "$@" 2>&1 | grep "ff_"
done
cp /tmp/libse_orig.so speech-enhancement_amd/libse_hip.so
