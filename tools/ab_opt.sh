#!/bin/bash
# same-box A/B of a library option on the conv shapes and on the pipeline: tools/ab_opt.sh "<key>=<a>" "<key>=<b>" ["<bench_conv filter>"]
A=$1; B=$2; F=${3:-.}
for O in $A $B $A $B; do
  echo "== CCNET_HIP_OPTIONS=$O"
  CCNET_HIP_OPTIONS=$O timeout -k 10 300 python3 tools/bench_conv.py --cold --iters 20 --filter "$F" 2>&1 | grep "us$" | awk '{printf "  %-34s %s us\n", substr($0,1,32), $(NF-1)}' || exit 1
done
[ -n "$NO_PIPE" ] && exit 0
for O in $A $B $A $B; do
  v=$(CCNET_HIP_OPTIONS=$O timeout -k 10 200 python3 bench.py --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-prof 2>&1 | grep -o '"value": [0-9.]*') || exit 1
  echo "pipeline batch 16, CCNET_HIP_OPTIONS=$O: $v"
done
