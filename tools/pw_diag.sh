#!/bin/bash
# which resource bounds the streaming pointwise kernel: variants built with -DPW_DIAG=<bits> (tools/build_variant.sh pwd<bits> "-DPW_DIAG=<bits>" gemm_conv)
F='L0 (proj|qkv|ff1\+GEGLU|o-proj)|L1 (qkv|ff1\+GEGLU)'
echo "== product"; timeout -k 10 120 python3 tools/bench_conv.py --filter "$F" --iters 20 2>&1 | grep -v "^ *!!" || exit 1
for v in 1 2 4 3 7; do
  echo "== PW_DIAG=$v (1 no stores, 2 no MFMAs, 4 no row fetch)"
  CCNET_HIP_LIB=$PWD/build_ab/lib_pwd$v.so timeout -k 10 120 python3 tools/bench_conv.py --filter "$F" --iters 20 2>&1 | grep -v "^ *!!" || exit 1
done
