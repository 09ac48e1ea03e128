#!/bin/bash
# whole-pipeline A/B of two builds of the library on one box (interleaved): tools/ab_lib.sh build_ab/lib_old.so ccnet_amd/libccnet_hip.so
for rep in 1 2; do for L in "$@"; do
  echo "== $L"
  CCNET_HIP_LIB=$PWD/$L timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --shapes gpurun_out/ab_shapes_$(basename $L .so)_$rep.txt 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernel_ms_total": [0-9.]*' | tr '\n' ' ' || exit 1
  echo
done; done
