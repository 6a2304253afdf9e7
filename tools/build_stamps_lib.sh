#!/bin/bash
# diagnostic variant of the library: the fused feed-forward backward with -DSE_FF_STAMPS, every other object from the product build
cd "$(dirname "$0")/.."
mkdir -p tools/micro/bin
objs=$(ls speech-enhancement_amd/build/*.hip.o | grep -v "/se_ff_fused")
for f in speech-enhancement_amd/csrc/se_ff_fused*.hip; do
  o=tools/micro/bin/$(basename $f .hip)_stamps.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -fPIC -std=c++17 -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops \
    -DSE_FF_STAMPS -c $f -o $o 2>/dev/null || exit 1
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/bin/libse_stamps.so $objs && echo built tools/micro/bin/libse_stamps.so
