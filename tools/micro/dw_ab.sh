for v in "" fb_NT4 fb_NO_B; do
  if [ -n "$v" ]; then export SE_HIP_LIB=$PWD/tools/micro/bin/libse_$v.so; else unset SE_HIP_LIB; fi
  echo "== ${v:-product}"; timeout -k 10 120 python tools/microbench.py dw_bench 2>/dev/null | grep "bwd fused\|dgrad+glu"
done
