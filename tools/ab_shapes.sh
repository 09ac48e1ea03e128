#!/bin/bash
# In-situ per-shape times (bench.py --shapes: eager pass, events around every launch) of library variants on one box:
#   tools/ab_shapes.sh "<shape regex>" lib1.so lib2.so ...      ("" = the product library)
pat=$1; shift
for L in "$@" "$@"; do
  if [ -n "$L" ] && [ "$L" != product ]; then export CCNET_HIP_LIB=$PWD/$L; else unset CCNET_HIP_LIB; fi
  out=gpurun_out/ab_shapes_$$.csv
  v=$(timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --shapes $out 2>/dev/null | grep -o '"value": [0-9.]*\|"kernel_ms_total": [0-9.]*' | tr '\n' ' ') || exit 1
  echo "== ${L:-product}: $v"
  grep -E "$pat" $out
done
