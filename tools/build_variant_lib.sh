#!/bin/bash
# variant of the library for a same-box A/B: tools/build_variant_lib.sh NAME "-DMACRO ..." file1.hip [file2.hip ...]
# -> tools/micro/bin/libse_NAME.so (the listed translation units rebuilt with the extra flags, every other object from the product build)
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
mkdir -p tools/micro/bin
objs=$(ls speech-enhancement_amd/build/*.hip.o)
for f in "$@"; do
  base=$(basename $f)
  pk="-Xclang -target-feature -Xclang -packed-fp32-ops"; [ "$base" = "se_dwconv.hip" ] && pk=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -fPIC -std=c++17 -Wno-unused-result $pk $flags \
    -c speech-enhancement_amd/csrc/$base -o tools/micro/bin/${base}_$name.o 2>/dev/null || exit 1
  objs=$(echo "$objs" | grep -v "/$base.o")
  objs="$objs
tools/micro/bin/${base}_$name.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/micro/bin/libse_$name.so $objs && echo built tools/micro/bin/libse_$name.so
