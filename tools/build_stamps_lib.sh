#!/bin/bash
# diagnostic variant of the library: se_ff_fused.hip with -DSE_FF_STAMPS, every other object from the product build
cd "$(dirname "$0")/.."
mkdir -p tools/micro/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -fPIC -std=c++17 -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops \
  -DSE_FF_STAMPS -c speech-enhancement_amd/csrc/se_ff_fused.hip -o tools/micro/bin/se_ff_fused_stamps.o 2>/dev/null || exit 1
objs=$(ls speech-enhancement_amd/build/*.hip.o | grep -v "/se_ff_fused.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/bin/libse_stamps.so $objs tools/micro/bin/se_ff_fused_stamps.o
