#!/bin/bash
# timing of diagnostic builds of the fused cross-attention kernel: tools/ab_xattn.sh lib_a.so lib_b.so ... ("" = product library)
for L in "$@"; do
  echo "== lib ${L:-product}"
  CCNET_HIP_LIB=$PWD/${L:-ccnet_amd/libccnet_hip.so} timeout -k 10 120 python3 tools/bench_xattn.py --iters 20 2>&1 | grep "^fused" || exit 1
done
