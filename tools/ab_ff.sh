#!/bin/bash
# same-box timing of fused feed-forward variants: tools/ab_ff.sh build_var/lib_a.so build_var/lib_b.so ... ("" = the product library)
for L in "$@"; do
  echo "== lib ${L:-product}"
  CCNET_HIP_LIB=$PWD/${L:-ccnet_amd/libccnet_hip.so} timeout -k 10 120 python3 tools/bench_ff.py --iters 20 2>&1 | grep "^fused" || exit 1
done
