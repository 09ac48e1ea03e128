#!/bin/bash
# same-box A/B of two builds of the library over batch sizes: tools/ab_batches.sh "1 2 4 8 16" (old = build_ab/lib_old.so, tools/build_ref.sh)
python3 bench.py --batch 2 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1  # clocks up
for B in ${1:-"1 2 4 16"}; do
  for L in old new old new; do
    lib=$PWD/ccnet_amd/libccnet_hip.so; [ $L = old ] && lib=$PWD/build_ab/lib_old.so
    v=$(CCNET_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*') || exit 1
    echo "batch $B $L $v"
  done
done
