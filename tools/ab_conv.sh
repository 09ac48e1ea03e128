#!/bin/bash
# same-box timing of conv kernel variants on selected shapes: tools/ab_conv.sh "<filter>" lib_a.so lib_b.so ... ("" = product library)
F=$1; shift
for L in "$@" "$@"; do
  echo "== lib ${L:-product}"
  CCNET_HIP_LIB=$PWD/${L:-ccnet_amd/libccnet_hip.so} timeout -k 10 200 python3 tools/bench_conv.py --cold --iters 20 --filter "$F" 2>&1 | grep "us$" | awk '{printf "  %-34s %s us\n", substr($0,1,32), $(NF-1)}' || exit 1
done
