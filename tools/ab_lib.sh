#!/bin/bash
# same-box A/B of two builds of the library on streaming-kernel shapes and on the pipeline: tools/ab_lib.sh build_ab/lib_base.so
F=${2:-'L0 (proj|qkv|ff1\+GEGLU|o-proj)|L1 (qkv|ff1\+GEGLU)'}
for L in $1 "" $1 ""; do
  echo "== lib ${L:-product}"
  CCNET_HIP_LIB=$PWD/${L:-ccnet_amd/libccnet_hip.so} timeout -k 10 120 python3 tools/bench_conv.py --filter "$F" --iters 20 2>&1 | grep "us$" | awk '{printf "  %-34s %s us\n", substr($0,1,32), $(NF-1)}' || exit 1
done
[ -n "$NO_PIPE" ] && exit 0
for L in $1 "" $1 ""; do
  v=$(CCNET_HIP_LIB=$PWD/${L:-ccnet_amd/libccnet_hip.so} timeout -k 10 200 python3 bench.py --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-prof 2>&1 | grep -o '"value": [0-9.]*') || exit 1
  echo "pipeline batch 16, lib ${L:-product}: $v"
done
