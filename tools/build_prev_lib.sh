#!/bin/bash
# the library of a git revision (default HEAD) for a same-box A/B against the working tree: tools/build_prev_lib.sh [rev] -> tools/micro/bin/libse_prev.so
rev=${1:-HEAD}
cd "$(dirname "$0")/.."
root=$(pwd)
rm -rf /tmp/se_prev && mkdir -p /tmp/se_prev/obj && git archive $rev speech-enhancement_amd/csrc include | tar -x -C /tmp/se_prev && cd /tmp/se_prev
for f in speech-enhancement_amd/csrc/*.hip; do
  b=$(basename $f); fl="-Xclang -target-feature -Xclang -packed-fp32-ops"; [ $b = se_dwconv.hip ] && fl=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -fPIC -std=c++17 -Wno-unused-result $fl -c $f -o obj/$b.o 2>/dev/null &
done
wait
mkdir -p $root/tools/micro/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/micro/bin/libse_prev.so obj/*.o && echo built tools/micro/bin/libse_prev.so from $rev
